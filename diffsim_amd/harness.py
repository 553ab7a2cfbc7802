"""Triplet / 2AFC benchmark harness with a per-(image, slot) feature cache.

The reference's drivers score every triplet with two full scorer calls, (A,B) and (A,C)
(``cute_main.py:111-132``, ``night_main.py:69-163``): image A is encoded, noised and pushed through
the U-Net twice with identical seed and slot, i.e. to bit-identical features.  Here the reference
image's features are computed once and shared by both pairs (3 U-Net forwards per triplet instead
of 4), every triplet of a chunk runs in one engine batch, and both scores of every triplet come
out of one fused tail launch (``dsim_pair_score`` takes index pairs into the feature tensor).

Decision rules mirror the reference:
  NIGHTS 2AFC  ``night_main.py:156-163``  cosine: predicted = 1 if s(ref,left) > s(ref,right) else 0;
                                          mse:    predicted = 1 if s(ref,left) < s(ref,right) else 0;
                                          correct when predicted == int(row['left_vote'])
  CUTE         ``cute_main.py:201-205``   correct when s(A,B) > s(A,C)  (B same instance, C other)
"""
from __future__ import annotations

import csv
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .diffsim import DiffSim, _norm_layer, get_generator
from .engine import pair_score
from .image import load_image, process_image
from .parallel import gather_scores, shard_triplets


@torch.no_grad()
def score_latent_triplets(scorer: DiffSim, lat_ref: torch.Tensor, lat_left: torch.Tensor, lat_right: torch.Tensor,
                          noiseA: torch.Tensor, noiseB: torch.Tensor, prompt, target_block="up_blocks", target_layer=0,
                          target_step=600, similarity="cosine", batch_triplets: int = 10, return_status: bool = False):
    """Scores (ref,left) and (ref,right) for every triplet; ref sits in slot A (noiseA), left and
    right in slot B (noiseB) exactly as two reference calls would place them.  Returns two (n,) f32
    device tensors that are bit-identical to 2n separate ``diffsim_latents`` calls."""
    n = lat_ref.shape[0]
    eng = scorer.engine(target_block, target_layer)
    s_l = torch.empty(n, dtype=torch.float32, device=scorer.device)
    s_r = torch.empty(n, dtype=torch.float32, device=scorer.device)
    bad = torch.zeros((), dtype=torch.int32, device=scorer.device)      # NaN guard: pairs whose score is not finite
    shp = lat_ref.shape[1:]
    for i0 in range(0, n, batch_triplets):
        i1 = min(n, i0 + batch_triplets)
        m = i1 - i0
        lat = torch.stack([lat_ref[i0:i1], lat_left[i0:i1], lat_right[i0:i1]], dim=1).reshape(3 * m, *shp)
        nz = torch.stack([noiseA.expand(m, *shp), noiseB.expand(m, *shp), noiseB.expand(m, *shp)], dim=1).reshape(3 * m, *shp)
        q, k, v = scorer.features(lat, nz, prompt, target_block, target_layer, target_step)
        base = torch.arange(0, 3 * m, 3, dtype=torch.int32, device=scorer.device)
        ia = torch.cat([base, base])
        ib = torch.cat([base + 1, base + 2])
        s, st = pair_score(q, k, v, ia, ib, eng.heads, similarity, return_status=True)
        bad += st.sum()
        s_l[i0:i1], s_r[i0:i1] = s[:m], s[m:]
    if return_status:
        return s_l, s_r, bad
    return s_l, s_r


def read_nights_csv(image_path: str, split: str = "val") -> List[dict]:
    """``data.csv`` of NIGHTS: split, ref_path, left_path, right_path, left_vote, prompt
    (night_main.py:53-67)."""
    rows = []
    with open(os.path.join(image_path, "data.csv"), mode="r") as f:
        for row in csv.DictReader(f):
            if row["split"] != split:
                continue
            rows.append({"ref": os.path.join(image_path, row["ref_path"]), "left": os.path.join(image_path, row["left_path"]),
                         "right": os.path.join(image_path, row["right_path"]), "vote": int(row["left_vote"]),
                         "prompt": f"An image of a {row['prompt'].lower()}"})
    return rows


def nights_decisions(s_left: torch.Tensor, s_right: torch.Tensor, similarity: str) -> torch.Tensor:
    if similarity == "mse":
        return (s_left < s_right).to(torch.int64)
    return (s_left > s_right).to(torch.int64)


def nights_accuracy(s_left, s_right, votes: Sequence[int], similarity: str = "cosine") -> float:
    pred = nights_decisions(s_left.cpu(), s_right.cpu(), similarity)
    v = torch.tensor(list(votes), dtype=torch.int64)
    return float((pred == v).float().mean() * 100.0) if len(v) else 0.0


def cute_accuracy(s_ab: torch.Tensor, s_ac: torch.Tensor) -> float:
    return float((s_ab.cpu() > s_ac.cpu()).float().mean() * 100.0) if s_ab.numel() else 0.0


@torch.no_grad()
def score_path_triplets(scorer, triplets: Sequence[Tuple[str, str, str, str]], img_size: int, target_block, target_layer,
                        target_step, seed=2333, similarity="cosine", rank: int = 0, world: int = 1, batch_triplets: int = 10):
    """Scores s(A,B) and s(A,C) of every (A, B, C, prompt) path triplet -- the two scorer calls per triplet the
    reference's loops make (cute_main.py:111-132, night_main.py:69-90) -- with whole triplets sharded over ranks (the
    cached reference-image features stay local), prompts encoded once each, images of a chunk encoded together, and
    one all_gather per side at the end.  Returns (s_ab, s_ac, n_nonfinite): length-len(triplets) f32 tensors on every
    rank and the number of NaN/inf pair scores (NaN guard)."""
    n = len(triplets)
    mine = shard_triplets(n, rank, world)
    dev = scorer.device
    if not isinstance(scorer, DiffSim):
        # DiffSim-XL / DiffSim-DiT scorers: their own diffsim_score per pair (same signature in the reference)
        sl, sr = [], []
        for j in mine:
            a, b, c, prompt = triplets[j]
            sl.append(scorer.diffsim_score(a, b, img_size, prompt, target_block, target_layer, target_step, similarity, seed))
            sr.append(scorer.diffsim_score(a, c, img_size, prompt, target_block, target_layer, target_step, similarity, seed))
        loc_l = torch.cat(sl).float().to(dev) if sl else torch.empty(0, dtype=torch.float32, device=dev)
        loc_r = torch.cat(sr).float().to(dev) if sr else torch.empty(0, dtype=torch.float32, device=dev)
        all_l, all_r = gather_scores(loc_l, n, rank, world), gather_scores(loc_r, n, rank, world)
        return all_l, all_r, int((~torch.isfinite(all_l)).sum() + (~torch.isfinite(all_r)).sum())
    layer = _norm_layer(target_layer)
    sl, sr = [], []
    nbad = torch.zeros((), dtype=torch.int32, device=dev)
    # prompts differ per row: group the shard by prompt so each context is encoded once
    by_prompt = {}
    for j in mine:
        by_prompt.setdefault(triplets[j][3], []).append(j)
    order, nA, nB = [], None, None
    vae = getattr(scorer, "vae", None)
    fast = vae is not None and hasattr(vae, "moments")        # HIP VAE: chunked encodes, threaded image decode
    eps = None
    nd = getattr(scorer, "noise_dtype", torch.float32)         # fp16 = the reference's fp16 pipeline draws
    for prompt, idxs in by_prompt.items():
        ref, left, right = [], [], []
        if fast:
            # every call reseeds the same generator: its draws (vae ref, vae other, noise ref, noise other) are the same
            # tensors for every triplet and for both (A,B) and (A,C)
            from .engine import _LatentDist
            if eps is None:
                g = get_generator(seed, "cpu")
                shp = None
            sf = vae.config.scaling_factor
            for c0 in range(0, len(idxs), batch_triplets):
                chunk = idxs[c0:c0 + batch_triplets]
                paths = [triplets[j][k] for j in chunk for k in (0, 1, 2)]
                ims = list(scorer._pool.map(lambda p_: process_image(load_image(p_), img_size), paths))
                d = _LatentDist(vae.moments(torch.cat(ims).to(vae.device).to(dtype=scorer.vae_dtype)))
                if eps is None:
                    shp = (1,) + tuple(d.mean.shape[1:])
                    eA = torch.randn(shp, generator=g, dtype=nd).float().to(vae.device)
                    eB = torch.randn(shp, generator=g, dtype=nd).float().to(vae.device)
                    nA = torch.randn(shp, generator=g, dtype=nd).float()
                    nB = torch.randn(shp, generator=g, dtype=nd).float()
                    eps = (eA, eB)
                ref.append((sf * (d.mean[0::3] + d.std[0::3] * eps[0])).to(nd).float())
                left.append((sf * (d.mean[1::3] + d.std[1::3] * eps[1])).to(nd).float())
                right.append((sf * (d.mean[2::3] + d.std[2::3] * eps[1])).to(nd).float())
        for j in ([] if fast else idxs):
            # one generator per (A,B) call; (A,C) reproduces the same A / noise draws
            pa, pb, pc, _ = triplets[j]
            g = get_generator(seed, "cpu")
            a = scorer.prepare_image_latents(process_image(load_image(pa), img_size), None, None, g)
            b = scorer.prepare_image_latents(process_image(load_image(pb), img_size), None, None, g)
            if nA is None:
                nA = torch.randn(a.shape, generator=g, dtype=nd).float()
                nB = torch.randn(a.shape, generator=g, dtype=nd).float()
            g2 = get_generator(seed, "cpu")
            scorer.prepare_image_latents(process_image(load_image(pa), img_size), None, None, g2)
            c = scorer.prepare_image_latents(process_image(load_image(pc), img_size), None, None, g2)
            ref.append(a.to(nd).float()); left.append(b.to(nd).float()); right.append(c.to(nd).float())
        a_, b_, bad = score_latent_triplets(scorer, torch.cat(ref), torch.cat(left), torch.cat(right), nA, nB, prompt,
                                            target_block, layer, target_step, similarity, batch_triplets, return_status=True)
        nbad += bad
        sl.append(a_); sr.append(b_); order += idxs
    if order:
        inv = torch.tensor(sorted(range(len(order)), key=lambda t: order[t]), dtype=torch.long, device=dev)
        loc_l, loc_r = torch.cat(sl)[inv], torch.cat(sr)[inv]
    else:
        loc_l = loc_r = torch.empty(0, dtype=torch.float32, device=dev)
    all_l = gather_scores(loc_l, n, rank, world)
    all_r = gather_scores(loc_r, n, rank, world)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(nbad)
    return all_l, all_r, int(nbad)


@torch.no_grad()
def nights_eval(scorer: DiffSim, image_path: str, img_size: int, target_block, target_layer, target_step, seed=2333,
                similarity="cosine", rank: int = 0, world: int = 1, batch_triplets: int = 10) -> float:
    """The whole night_main.py loop (csv -> triplets -> 2AFC accuracy), triplets sharded over ranks."""
    rows = read_nights_csv(image_path)
    trip = [(r["ref"], r["left"], r["right"], r["prompt"]) for r in rows]
    all_l, all_r, _bad = score_path_triplets(scorer, trip, img_size, target_block, target_layer, target_step, seed, similarity,
                                             rank, world, batch_triplets)
    return nights_accuracy(all_l, all_r, [r["vote"] for r in rows], similarity)
