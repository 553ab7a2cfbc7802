#!/usr/bin/env python3
"""Development probe: do two independent half-batches on two HIP streams overlap (one stream's HBM-bound kernels --
norms, short-K linears -- under the other's MFMA-bound convolutions / attention)?  Same kernels, same results; only the
enqueue order changes.  Prints pairs/s for 1 stream x 2 sequential calls and for 2 concurrent streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsim_amd import config as C, scheduler as sched, synth as S      # noqa: E402
from diffsim_amd.engine import UNetEngine, pair_score                    # noqa: E402

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = C.SD15
keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out"))]
sd = S.make_state_dict(cfg, seed=0, keys=keys)
dev = torch.device("cuda:0")
engs = [UNetEngine(cfg, sd, torch.bfloat16, "up_blocks", 0, "cuda:0") for _ in range(2)]
t = sched.timestep_from_index(600)
sa, sb = sched.noise_coefficients(t)
for e in engs:
    e.set_timestep(t)
noise = S.draw_pair_noise(2334, (1, 4, 64, 64))
lat = [torch.cat([torch.cat(S.make_pair_latents(cfg, h * bp + i)) for i in range(bp)]).to(dev) for h in range(2)]
nz = torch.cat([noise[2], noise[3]] * bp).to(dev)
ctx = S.make_context(cfg).to(dev)
ia = torch.arange(0, 2 * bp, 2, dtype=torch.int32, device=dev)
shape = (2 * bp, 2, engs[0].tokens, engs[0].heads * engs[0].head_dim)
outs = [tuple(torch.empty(shape, dtype=torch.bfloat16, device=dev) for _ in range(3)) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def half(h, stream):
    with torch.cuda.stream(stream):
        q, k, v = engs[h].qkv(lat[h], nz, sa, sb, ctx, out=outs[h])
        return pair_score(q, k, v, ia, ia + 1, engs[h].heads, "cosine")


def run(concurrent, steps=8):
    for _ in range(2):
        half(0, streams[0]); half(1, streams[1 if concurrent else 0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s0 = half(0, streams[0])
        s1 = half(1, streams[1 if concurrent else 0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return steps * 2 * bp / dt, s0, s1


for mode in (False, True, False, True):
    r, s0, s1 = run(mode)
    print(f"{'2 streams' if mode else '1 stream '}  {r:8.1f} pairs/s   scores {float(s0[0]):.6f} {float(s1[0]):.6f}", flush=True)
