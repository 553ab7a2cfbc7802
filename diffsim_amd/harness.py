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
from .image import DecodePool, load_image, process_image
from .parallel import gather_scores, shard_triplets


@torch.no_grad()
def score_latent_triplets(scorer, lat_ref: torch.Tensor, lat_left: torch.Tensor, lat_right: torch.Tensor,
                          noiseA: torch.Tensor, noiseB: torch.Tensor, prompt, target_block="up_blocks", target_layer=0,
                          target_step=600, similarity="cosine", batch_triplets: Optional[int] = None, return_status: bool = False):
    """Scores (ref,left) and (ref,right) for every triplet; ref sits in slot A (noiseA), left and
    right in slot B (noiseB) exactly as two reference calls would place them.  Returns two (n,) f32
    device tensors that are bit-identical to 2n separate ``diffsim_latents`` calls (any scorer kind).
    batch_triplets=None: the engine batch of the measured optimum (``_Adapter.auto_triplets``)."""
    s_l, s_r, bad = _score_chunks(_Adapter(scorer), lat_ref, lat_left, lat_right, noiseA, noiseB, prompt, target_block,
                                  target_layer, target_step, similarity, batch_triplets)
    return (s_l, s_r, bad) if return_status else (s_l, s_r)


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


class _Adapter:
    """What the batched triplet path needs to know about a scorer kind -- the per-call arithmetic of its reference entry
    point, stated once: DiffSim.diffsim (diffsim/diffsim.py:98-197), diffsim_xl.diffsim_score (diffsim_xl.py:65-155),
    diffsim_DiT.diffsim_score (diffsim_dit.py:74-142).  All three reseed one generator per call and draw, in this order,
    the VAE sample of image A, of image B, the noise of A, of B -- so the four draws are the same tensors for every pair."""

    def __init__(self, scorer):
        from .diffsim_dit import diffsim_DiT
        from .diffsim_xl import diffsim_xl
        self.s = scorer
        self.kind = "sd15" if isinstance(scorer, DiffSim) else ("xl" if isinstance(scorer, diffsim_xl) else
                                                                ("dit" if isinstance(scorer, diffsim_DiT) else None))
        if self.kind is None:
            raise TypeError(f"no triplet adapter for {type(scorer).__name__}")
        nd = getattr(scorer, "noise_dtype", torch.float32)
        vae = getattr(scorer, "vae", None)
        self.vae = vae
        self.fast = vae is not None and hasattr(vae, "moments")      # the HIP VAE encoder: chunked, look-ahead decode
        if self.kind == "sd15":
            self.image_half = scorer.vae_dtype == torch.float16      # image.to(dtype=float16), diffsim.py:93
            self.eps_dtype = nd                                      # latent_dist.sample draws in the pipeline dtype
            self.noise_draw = nd
            self.round16 = nd == torch.float16
        else:
            self.image_half = False                                  # the SDXL / DiT VAE runs in fp32 (diffsim_xl.py:61)
            self.eps_dtype = getattr(vae, "sample_dtype", torch.float32)
            self.noise_draw = nd if self.kind == "xl" else torch.float16      # DiT: randn_tensor(dtype=latents.dtype) = fp16
            self.round16 = True                                      # latents.to(dtype=float16), diffsim_xl.py:63 / diffsim_dit.py:59
        self._ctx = {}

    def group_key(self, prompt):
        return None if self.kind == "dit" else prompt                # DiT ignores the prompt (labels [1, 1000])

    def heads(self, block, layer):
        if self.kind == "dit":
            return self.s.engine(int(layer[0])).heads
        return self.s.engine(block, layer if self.kind == "xl" else _norm_layer(layer)).heads

    def engine(self, block, layer):
        if self.kind == "dit":
            return self.s.engine(int(layer[0]))
        return self.s.engine(block, layer if self.kind == "xl" else _norm_layer(layer))

    def auto_triplets(self, block, layer, n: int) -> int:
        """Triplets per engine batch when the caller names none: the image count of the batch sweeps' optimum (SD1.5 and
        DiT: 128 images = 64 pairs, profiles/r04h_batch_sweep.txt; SDXL at 1024 px: 16), inside the 2 GiB activation bound
        and half of the free HBM."""
        eng = self.engine(block, layer)
        t = max(1, min((16 if self.kind == "xl" else 128) // 3, max(1, int(n))))
        if hasattr(eng, "max_images"):
            t = max(1, min(t, eng.max_images() // 3))
        try:
            free, _total = torch.cuda.mem_get_info(self.s.device)
            while t > 1 and hasattr(eng, "workspace_bytes") and eng.workspace_bytes(3 * t) > 0.5 * free:
                t = (t + 1) // 2
        except Exception:
            pass
        return t

    def features(self, lat, nz, prompt, block, layer, step):
        if self.kind == "sd15":
            return self.s.features(lat, nz, prompt, block, _norm_layer(layer), step)
        if self.kind == "xl":
            if prompt not in self._ctx:
                if self.s._encode_prompt is None:
                    raise RuntimeError("no text encoder plugged in: pass encode_prompt=...")
                self._ctx[prompt] = self.s._encode_prompt(prompt)    # (context, pooled): once per prompt, not once per pair
            ctx, pooled = self._ctx[prompt]
            return self.s.features(lat, nz, ctx, pooled, block, layer, step)
        return self.s.features(lat, nz, int(layer[0]), step)


@torch.no_grad()
def _score_chunks(ad: _Adapter, ref, left, right, nA, nB, prompt, block, layer, step, similarity, batch_triplets):
    """(ref,left) and (ref,right) scores of latent triplets: 3 forwards per triplet (the reference image's features are
    shared), chunked engine batches, one fused tail launch per chunk."""
    n = ref.shape[0]
    dev = ad.s.device
    s_l = torch.empty(n, dtype=torch.float32, device=dev)
    s_r = torch.empty(n, dtype=torch.float32, device=dev)
    bad = torch.zeros((), dtype=torch.int32, device=dev)
    shp = ref.shape[1:]
    heads = ad.heads(block, layer)
    if batch_triplets is None:
        batch_triplets = ad.auto_triplets(block, layer, n)
    for i0 in range(0, n, batch_triplets):
        i1 = min(n, i0 + batch_triplets)
        m = i1 - i0
        lat = torch.stack([ref[i0:i1], left[i0:i1], right[i0:i1]], dim=1).reshape(3 * m, *shp)
        nz = torch.stack([nA.expand(m, *shp), nB.expand(m, *shp), nB.expand(m, *shp)], dim=1).reshape(3 * m, *shp)
        q, k, v = ad.features(lat, nz, prompt, block, layer, step)
        base = torch.arange(0, 3 * m, 3, dtype=torch.int32, device=dev)
        s, st = pair_score(q, k, v, torch.cat([base, base]), torch.cat([base + 1, base + 2]), heads, similarity, return_status=True)
        bad += st.sum()
        s_l[i0:i1], s_r[i0:i1] = s[:m], s[m:]
    return s_l, s_r, bad


@torch.no_grad()
def score_path_triplets(scorer, triplets: Sequence[Tuple[str, str, str, str]], img_size: int, target_block, target_layer,
                        target_step, seed=2333, similarity="cosine", rank: int = 0, world: int = 1, batch_triplets: int = 10,
                        unet_triplets: Optional[int] = None):
    """Scores s(A,B) and s(A,C) of every (A, B, C, prompt) path triplet -- the two scorer calls per triplet the
    reference's loops make (cute_main.py:111-132, night_main.py:69-90, style_main.py:150-175) -- for all three scorer
    kinds (DiffSim, diffsim_xl, diffsim_DiT): whole triplets sharded over ranks (the cached reference-image features stay
    local), prompts encoded once each, 3 forwards per triplet instead of 4, images of a chunk encoded together.

    Host work is only decode + Lanczos resize, on a thread pool that runs two chunks ahead of the GPU; the /255,
    (x-0.5)/0.5, NCHW and fp16-cast arithmetic of process_image and the posterior sampling run on the device
    (dsim_image_preprocess / dsim_latent_sample), bit-identically to the per-pair path.  Returns (s_ab, s_ac, n_nonfinite):
    length-len(triplets) f32 tensors on every rank and the number of NaN/inf pair scores (NaN guard)."""
    from .engine import image_preprocess, latent_sample
    n = len(triplets)
    mine = shard_triplets(n, rank, world)
    ad = _Adapter(scorer)
    dev = scorer.device
    sl, sr, order = [], [], []
    nbad = torch.zeros((), dtype=torch.int32, device=dev)
    # prompts differ per row: group the shard by prompt so each context is encoded once
    groups = {}
    for j in mine:
        groups.setdefault(ad.group_key(triplets[j][3]), []).append(j)
    draws = None            # (eA, eB on the device; nA, nB): the same four tensors for every triplet (one reseeded generator)
    pool = getattr(scorer, "_decode", None) or _shared_pool()
    for key, idxs in groups.items():
        prompt = triplets[idxs[0]][3]
        ref, left, right = [], [], []
        if ad.fast:
            vae = ad.vae
            sf = vae.config.scaling_factor
            chunks = [idxs[c0:c0 + batch_triplets] for c0 in range(0, len(idxs), batch_triplets)]

            def submit(chunk):
                return pool.submit([triplets[j][k] for j in chunk for k in (0, 1, 2)], img_size)
            pending = [submit(c) for c in chunks[:2]]                # decode runs two chunks ahead of the GPU
            for ci, chunk in enumerate(chunks):
                px = DecodePool.gather(pending.pop(0))
                if ci + 2 < len(chunks):
                    pending.append(submit(chunks[ci + 2]))
                x = image_preprocess(px.to(vae.device, non_blocking=True), ad.image_half)
                mom = vae.moments(x)
                if draws is None:
                    g = get_generator(seed, "cpu")
                    shp = (1, mom.shape[1] // 2) + tuple(mom.shape[2:])
                    eA = torch.randn(shp, generator=g, dtype=ad.eps_dtype).float().to(vae.device)
                    eB = torch.randn(shp, generator=g, dtype=ad.eps_dtype).float().to(vae.device)
                    nA = torch.randn(shp, generator=g, dtype=ad.noise_draw).float()
                    nB = torch.randn(shp, generator=g, dtype=ad.noise_draw).float()
                    draws = (eA, eB, nA, nB)
                ref.append(latent_sample(mom, draws[0], sf, 0, 3, ad.round16))
                left.append(latent_sample(mom, draws[1], sf, 1, 3, ad.round16))
                right.append(latent_sample(mom, draws[1], sf, 2, 3, ad.round16))
        else:
            # no HIP VAE plugged in: the scorer's own prepare_image_latents per image, in the reference's draw order
            for j in idxs:
                pa, pb, pc, _ = triplets[j]
                g = get_generator(seed, "cpu")
                a = _prepare(scorer, ad, process_image(load_image(pa), img_size), g)
                b = _prepare(scorer, ad, process_image(load_image(pb), img_size), g)
                if draws is None:
                    nA = torch.randn(a.shape, generator=g, dtype=ad.noise_draw).float()
                    nB = torch.randn(a.shape, generator=g, dtype=ad.noise_draw).float()
                    draws = (None, None, nA, nB)
                g2 = get_generator(seed, "cpu")                      # the (A, C) call: same A draw, then C's
                _prepare(scorer, ad, process_image(load_image(pa), img_size), g2)
                c = _prepare(scorer, ad, process_image(load_image(pc), img_size), g2)
                ref.append(a); left.append(b); right.append(c)
        # (batch_triplets sizes the decode / VAE-encode chunks above; the U-Net batch is chosen by the adapter)
        a_, b_, bad = _score_chunks(ad, torch.cat(ref), torch.cat(left), torch.cat(right), draws[2], draws[3], prompt,
                                    target_block, target_layer, target_step, similarity, unet_triplets)
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


def _prepare(scorer, ad: _Adapter, tensor, generator):
    """prepare_image_latents of the scorer kind, returned as the f32 values its pipeline carries on."""
    if ad.kind == "sd15":
        return scorer.prepare_image_latents(tensor, None, None, generator).to(ad.noise_draw).float()
    return scorer.prepare_image_latents(tensor, generator).float()


_POOL = None


def _shared_pool():
    """Decode pool for scorers that own none (diffsim_xl, diffsim_DiT): the host's cores divided among the node's ranks."""
    global _POOL
    if _POOL is None:
        _POOL = DecodePool()
    return _POOL


@torch.no_grad()
def nights_eval(scorer, image_path: str, img_size: int, target_block, target_layer, target_step, seed=2333,
                similarity="cosine", rank: int = 0, world: int = 1, batch_triplets: int = 10) -> float:
    """The whole night_main.py loop (csv -> triplets -> 2AFC accuracy), triplets sharded over ranks."""
    rows = read_nights_csv(image_path)
    trip = [(r["ref"], r["left"], r["right"], r["prompt"]) for r in rows]
    all_l, all_r, _bad = score_path_triplets(scorer, trip, img_size, target_block, target_layer, target_step, seed, similarity,
                                             rank, world, batch_triplets)
    return nights_accuracy(all_l, all_r, [r["vote"] for r in rows], similarity)
