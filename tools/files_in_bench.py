#!/usr/bin/env python3
"""Development probe: files-in throughput (PNG decode + Lanczos resize on the host thread pool, VAE + text encoder +
U-Net + tail on the GPU), synthetic weights, random 600x500 PNGs.  Two legs, one JSON line:
  pairs     DiffSim.score_pairs over 64 path pairs with one prompt (the CUTE-style call pattern)
  triplets  harness.score_path_triplets over 192 (ref, left, right, prompt) rows with 12 distinct per-row prompts, each
            encoded once by the CLIP-L-sized text encoder on the device (the NIGHTS call pattern, night_main.py:59-90)"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsim_amd import config as C, harness as H, synth as S, text as T        # noqa: E402
from diffsim_amd.diffsim import DiffSim                                          # noqa: E402
from diffsim_amd.engine import VAEEncoder                                        # noqa: E402

d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
paths = []
for i in range(32):
    p = os.path.join(d, f"im{i}.png")
    Image.fromarray(rng.integers(0, 255, (500, 600, 3), dtype=np.uint8)).save(p)
    paths.append(p)
NP = int(os.environ.get("FILES_IN_PAIRS", "256"))
pairs = [(paths[i % 32], paths[(i * 7 + 3) % 32]) for i in range(NP)]
vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), torch.bfloat16, "cuda")
g = torch.Generator().manual_seed(1)
tsd = {k: (0.02 * torch.randn(s, generator=g) if not (k.endswith("weight") and "norm" in k) else 1.0 + 0.02 * torch.randn(s, generator=g))
       for k, s in T.clip_text_param_shapes(T.CLIP_L).items()}
enc = T.CLIPTextEncoder(T.CLIP_L, tsd, device="cuda")
n_enc = [0]


def tokenize(p):          # stand-in tokenizer (the CLIP vocabulary files are not available offline): 77 ids, EOS-padded
    n_enc[0] += 1
    ids = [49406] + [1000 + (ord(c) * 31) % 40000 for c in p][:75]
    return torch.tensor([ids + [49407] * (77 - len(ids))])


ds = DiffSim(torch.bfloat16, "cuda", state_dict=S.make_state_dict(C.SD15, seed=0), vae=vae,
             encode_prompt=T.make_encode_prompt(enc, tokenize),
             decode_procs=int(os.environ["FILES_IN_PROCS"]) if "FILES_IN_PROCS" in os.environ else None)
ds.score_pairs(pairs[:16], 512, "a photo", "up_blocks", 0, 600, seed=2334, batch_pairs=16)
torch.cuda.synchronize(); t0 = time.perf_counter()
s = ds.score_pairs(pairs, 512, "a photo", "up_blocks", 0, 600, seed=2334, batch_pairs=16)
torch.cuda.synchronize(); dt_pairs = time.perf_counter() - t0

# ---- per-stage times of the same work ----------------------------------------------------------------------------
from diffsim_amd.engine import image_preprocess, latent_sample          # noqa: E402
from diffsim_amd.image import host_threads, load_image, resize_u8       # noqa: E402
t0 = time.perf_counter()
for p_ in paths[:8]:
    resize_u8(load_image(p_), 512)
host_ms_per_image = (time.perf_counter() - t0) / 8 * 1e3                # one thread: PNG decode + EXIF + Lanczos resize
t0 = time.perf_counter()
from diffsim_amd.image import DecodePool                                # noqa: E402
px = DecodePool.gather(ds._decode.submit([p for ab in pairs[:128] for p in ab], 512))
pool_images_per_s = 256 / (time.perf_counter() - t0)                    # the decode pool alone, no GPU work beside it
pxd = px[:32].cuda()
eps = torch.randn(1, 4, 64, 64, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4):
    mom = vae.moments(image_preprocess(pxd, True))
    lat = latent_sample(mom, eps, 0.18215, 0, 2)
torch.cuda.synchronize(); vae_ms_per_pair = (time.perf_counter() - t0) / 4 / 16 * 1e3
latA, latB = lat, latent_sample(mom, eps, 0.18215, 1, 2)
nz = torch.randn(1, 4, 64, 64)
ds.score_latent_pairs(latA, latB, nz, nz, "a photo", "up_blocks", 0, 600, "cosine", batch_pairs=16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4):
    ds.score_latent_pairs(latA, latB, nz, nz, "a photo", "up_blocks", 0, 600, "cosine", batch_pairs=16)
torch.cuda.synchronize(); unet_ms_per_pair = (time.perf_counter() - t0) / 4 / 16 * 1e3

trip = [(paths[i % 32], paths[(i * 5 + 1) % 32], paths[(i * 11 + 2) % 32], f"An image of a thing number {i % 12}") for i in range(192)]
H.score_path_triplets(ds, trip[:12], 512, "up_blocks", [0], 600, 2334, "cosine", batch_triplets=10)
ds._ctx.clear()
n_enc[0] = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
sl, sr, bad = H.score_path_triplets(ds, trip, 512, "up_blocks", [0], 600, 2334, "cosine", batch_triplets=10)
torch.cuda.synchronize(); dt_trip = time.perf_counter() - t0
prompt_encodes = n_enc[0] // 2                     # two tokenisations (negative, positive) per encoded prompt
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20):
    enc(tokenize(f"prompt {i}"))
torch.cuda.synchronize(); dt_txt = (time.perf_counter() - t0) / 20
print(json.dumps({"probe": "files-in (decode + resize + VAE + text + U-Net + tail), SD1.5 512 px, bf16, synthetic weights",
                  "pairs_per_s_one_prompt": round(len(pairs) / dt_pairs, 1), "pairs": len(pairs),
                  "triplets_per_s_per_row_prompts": round(len(trip) / dt_trip, 1), "pair_scores_per_s_triplets": round(2 * len(trip) / dt_trip, 1),
                  "triplets": len(trip), "distinct_prompts": 12, "prompt_encodes": prompt_encodes, "nonfinite_scores": bad,
                  "clip_l_text_encoder_ms_per_prompt": round(1e3 * dt_txt, 2), "host_cpu_count": os.cpu_count(),
                  "stages": {"host_decode_resize_ms_per_image_one_thread": round(host_ms_per_image, 2), "decode_workers": ("%d processes" % ds._decode.procs) if ds._decode.procs > 0 else ("%d threads" % host_threads()),
                             "pool_images_per_s": round(pool_images_per_s, 1),
                             "gpu_preprocess_vae_sample_ms_per_pair": round(vae_ms_per_pair, 3),
                             "gpu_unet_tail_ms_per_pair_batch16": round(unet_ms_per_pair, 3),
                             "gpu_bound_pairs_per_s": round(1e3 / (vae_ms_per_pair + unet_ms_per_pair), 1)},
                  "score_sample": [round(float(x), 5) for x in s[:3]]}))
