#!/usr/bin/env python3
"""The tap layer alone, for rocprofv3 --pmc passes (north_star: MFMA utilisation on the up_blocks[0] attention GEMM):
the tapped to_q | to_k | to_v projection at the bench batch as ONE 65536 x 3840 x 1280 bf16 launch through op_linear (the same
gemm_kernel instantiation, tile walk and K loop as the engine's DSIM_FUSE_TAPQKV launch; the engine's launch additionally
splits its output columns over three tensors in the epilogue -- GemmArgs.out_split, a per-tile store-descriptor select -- which
op_linear does not exercise; up to round 3 the engine made three 1280-column launches) and the fused score tail (4 x SDPA of 256 tokens x 8 heads x 160
+ cosine) for 64 pairs.  tools/profile_round.sh runs it under the counters and
profiles/summarize_tap.py turns the csv into profiles/<tag>_tap_pmc.json."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsim_amd import engine as E          # noqa: E402


def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    M, C = 256 * 256, 1280
    x = torch.randn(M, C, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dev)
    for _ in range(6):
        E.op_linear(x, w)
    q, k, v = (torch.randn(128, 2, 256, C, generator=g).to(dev, torch.bfloat16) for _ in range(3))
    ia = torch.arange(0, 128, 2, dtype=torch.int32, device=dev)
    for _ in range(6):
        s = E.pair_score(q, k, v, ia, ia + 1, 8, "cosine")
    torch.cuda.synchronize()
    print("tap probe done", float(s[0]))


if __name__ == "__main__":
    main()
