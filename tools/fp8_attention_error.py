import sys, math, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from diffsim_amd import engine as E
for (B,H,N,D) in ((3,16,256,72),(2,4,64,32)):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B,N,H*D,generator=g)*0.9; k = torch.randn(B,N,H*D,generator=g)*0.9; v = torch.randn(B,N,H*D,generator=g)
    qh,kh,vh = (t.bfloat16().float() for t in (q,k,v))
    want = F.scaled_dot_product_attention(qh.view(B,N,H,D).transpose(1,2), kh.view(B,N,H,D).transpose(1,2), vh.view(B,N,H,D).transpose(1,2)).transpose(1,2).reshape(B,N,H*D)
    qd,kd,vd = (t.cuda().bfloat16().contiguous() for t in (q,k,v))
    g8 = E.op_attention(qd,kd,vd,H,fp8=True).float().cpu(); g16 = E.op_attention(qd,kd,vd,H).float().cpu()
    sc = float(want.abs().max())
    print(D, "fp8 max/mean rel", float((g8-want).abs().max())/sc, float((g8-want).abs().mean())/sc, " bf16 max rel", float((g16-want).abs().max())/sc)
