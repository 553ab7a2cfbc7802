"""Development probe: the 3x3 conv GEMM on an SD1.5 level-0 / level-2 shape (for rocprofv3 --pmc passes)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import engine as E

B, H, W, Cin, Cout = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (128, 16, 16, 2560, 1280)))
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, H, W, Cin, generator=g)).to("cuda", torch.bfloat16)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).cuda()
b = torch.randn(Cout, generator=g).cuda()
for _ in range(3):
    o = E.op_conv3x3(x, w, b)
torch.cuda.synchronize()
print(float(o.float().abs().mean()))
