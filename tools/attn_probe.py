"""Development probe: run the self-attention kernel on the SD1.5 level-0 shape (for rocprofv3 --pmc passes)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import engine as E

B, H, N, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 8, 4096, 40)))
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, N, 3 * H * D, generator=g) * 0.5).to("cuda", torch.bfloat16)
q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].contiguous() for i in range(3))
for _ in range(3):
    o = E.op_attention(q, k, v, H)
torch.cuda.synchronize()
print(float(o.float().abs().mean()))
