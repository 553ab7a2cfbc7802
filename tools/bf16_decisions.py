#!/usr/bin/env python3
"""2AFC decision agreement of the 16-bit paths (bf16, fp16) with the fp32 kernel mode (SURVEY.md section 8d parity gate: "bf16 mode reports
max/mean error and 2AFC decision flips").  The fp32 mode is the one pinned to the CPU oracle at 1e-4 (tests/test_gpu_fullsize.py);
this tool measures what the bf16 and the fp16 kernels do to the DECISIONS of a NIGHTS-style run (night_main.py:156-163: left wins when
s(ref,left) > s(ref,right)): N synthetic full-size triplets (SD1.5, 512 px latents, up_blocks[0], step 600, cosine), the two
distortions of a triplet drawn at nearby strengths so that a good share of the margins is small.

    python tools/bf16_decisions.py [N=1000] > profiles/r04_decisions_16bit.json
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsim_amd import config as C, harness as H, synth as S      # noqa: E402
from diffsim_amd.diffsim import DiffSim                              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = C.SD15
keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out",
                                                                  "up_blocks.1.attentions.2.transformer_blocks.0.attn2",
                                                                  "up_blocks.1.attentions.2.transformer_blocks.0.ff",
                                                                  "up_blocks.1.attentions.2.proj_out", "up_blocks.1.upsamplers"))]
sd = S.make_state_dict(cfg, seed=0, keys=keys)
ctx = S.make_context(cfg)
g = torch.Generator().manual_seed(20260301)
shp = (N, 4, 64, 64)
ref = torch.randn(shp, generator=g) * 0.9
# left / right: the reference mixed with independent noise at strengths a, b in [0.15, 0.85], |a - b| small for most triplets
a = 0.15 + 0.7 * torch.rand(N, 1, 1, 1, generator=g)
b = (a + 0.08 * torch.randn(N, 1, 1, 1, generator=g)).clamp(0.05, 0.95)
left = (1 - a * a).sqrt() * ref + a * torch.randn(shp, generator=g) * 0.9
right = (1 - b * b).sqrt() * ref + b * torch.randn(shp, generator=g) * 0.9
n = S.draw_pair_noise(2334, (1, 4, 64, 64))
out = {}
for name, dt, bt in (("fp32", torch.float32, 8), ("bf16", torch.bfloat16, 20), ("fp16", torch.float16, 20)):
    ds = DiffSim(torch_dtype=dt, device="cuda", unet_config=cfg, state_dict=sd)
    t0 = time.perf_counter()
    sl, sr = H.score_latent_triplets(ds, ref.cuda(), left.cuda(), right.cuda(), n[2], n[3], ctx, "up_blocks", 0, 600, "cosine", bt)
    torch.cuda.synchronize()
    out[name] = (sl.double().cpu(), sr.double().cpu(), time.perf_counter() - t0)
    del ds
    torch.cuda.empty_cache()
fl, fr, tf = out["fp32"]
edges = [0.0, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 1.0]
m32 = fl - fr                                         # decision margins of the fp32 kernel mode
res = {"probe": "2AFC decisions, bf16 and fp16 kernels vs fp32 kernel mode, SD1.5 512 px latents-in, up_blocks[0] step 600 cosine, synthetic weights",
       "triplets": N, "margin_abs_edges": edges,
       "triplets_per_margin_bin": [int(((m32.abs() >= lo) & (m32.abs() < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])],
       "fp32_margin_abs_median": float(m32.abs().median()), "fp32_seconds": round(tf, 1), "fp32_scores_sample": fl[:4].tolist()}
for name in ("bf16", "fp16"):
    bl, br, tb = out[name]
    m16 = bl - br
    flips = ((m32 > 0) != (m16 > 0))
    dscore = torch.cat([(fl - bl).abs(), (fr - br).abs()])
    res[name] = {"decision_flips": int(flips.sum()), "flip_rate": float(flips.float().mean()),
                 "largest_fp32_margin_among_flips": float(m32[flips].abs().max()) if flips.any() else 0.0,
                 "flips_per_margin_bin": [int(((m32.abs() >= lo) & (m32.abs() < hi) & flips).sum()) for lo, hi in zip(edges[:-1], edges[1:])],
                 "score_abs_diff_max": float(dscore.max()), "score_abs_diff_mean": float(dscore.mean()), "seconds": round(tb, 1)}
print(json.dumps(res))
