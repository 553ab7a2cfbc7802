"""Batch-size sweep of the headline path: pairs/step in {1,2,4,8,16,32,48,64}, latents-in, bf16.
Prints one JSON line per batch size.  Run on the GPU box: python tools/sweep_batch.py"""
import json
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim

cfg = C.SD15
ds = DiffSim(torch.bfloat16, "cuda", state_dict=S.make_state_dict(cfg, seed=0))
ctx = S.make_context(cfg).cuda()
n = [t.cuda() for t in S.draw_pair_noise(2334, (1, 4, 64, 64))]
for bp in (1, 2, 4, 8, 16, 32, 48, 64):        # every activation stays < 2 GiB (32-bit buffer offsets) up to 68 pairs
    prs = [S.make_pair_latents(cfg, i) for i in range(bp)]
    zA, zB = torch.cat([p[0] for p in prs]).cuda(), torch.cat([p[1] for p in prs]).cuda()
    run = lambda: ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, "up_blocks", 0, 600, "cosine", batch_pairs=bp)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    steps = max(4, 64 // bp)
    t0 = time.perf_counter()
    for _ in range(steps):
        s = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"pairs_per_step": bp, "ms_per_step": round(dt * 1e3, 3), "pairs_per_s": round(bp / dt, 1),
                      "tflops": round(bp * 1580.4e9 / dt / 1e12, 1)}), flush=True)
