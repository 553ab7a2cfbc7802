"""Development probe: a linear GEMM (default: the tapped layer's fused q/k/v projection at 32 pairs) for --pmc passes.
usage: lin_probe.py M N K [geglu] [res]"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import engine as E

M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 3840, 1280)))
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to("cuda", torch.bfloat16)
w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
geglu = "geglu" in sys.argv
bias = torch.randn(N, generator=g).cuda()
res = torch.randn(M, N, generator=g).to("cuda", torch.bfloat16) if "res" in sys.argv else None
for _ in range(3):
    o = E.op_linear(x, w, bias=bias, residual=res, geglu=geglu)
torch.cuda.synchronize()
print(float(o.float().abs().mean()))
