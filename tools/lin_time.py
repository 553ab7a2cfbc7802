"""Development probe: ablation timing of linear GEMM shapes (HIP events over 20 launches) under the DSIM_DBG hooks
(1 no MFMA, 2 stores dropped, 4 residual loads short-circuited, 8 A loads short-circuited, 16 W loads short-circuited).
The hooks are NOT in the tree: apply the patch at the end of profiles/r02_linear_ablation.txt to gemm.hip / common.h first
(without it every mask times the unmodified kernel)."""
import os
import sys
import torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import engine as E

SHAPES = [(524288, 320, 320, False, True), (524288, 320, 320, False, False), (524288, 960, 320, False, False),
          (131072, 640, 640, False, True), (524288, 2560, 320, True, False), (32768, 1280, 1280, False, True)]
DBG = [0, 1, 2, 4, 8, 6, 14, 15, 30, 31]
g = torch.Generator().manual_seed(0)
for M, N, K, geglu, has_res in SHAPES:
    x = torch.randn(M, K, generator=g).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    No = N // 2 if geglu else N
    res = torch.randn(M, No, generator=g).to("cuda", torch.bfloat16) if has_res else None
    line = []
    for d in DBG:
        os.environ["DSIM_DBG"] = str(d)
        for _ in range(3):
            o = E.op_linear(x, w, bias=bias, residual=res, geglu=geglu)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            o = E.op_linear(x, w, bias=bias, residual=res, geglu=geglu)
        b.record()
        torch.cuda.synchronize()
        line.append("%d:%.0f" % (d, a.elapsed_time(b) / 20 * 1e3))
    print(f"M{M} N{N} K{K} geglu={int(geglu)} res={int(has_res)} us/launch by dbg mask: " + "  ".join(line), flush=True)
    del x, w, res, o
