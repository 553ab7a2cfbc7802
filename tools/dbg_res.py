import torch, math, sys
sys.path.insert(0, '.')
import torch.nn.functional as F
from diffsim_amd import engine as eng
def run(M,N,K,dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    q=lambda t: t.to(dtype).float()
    outs=[]
    for rep in range(3):
        got = eng.op_linear(x.to(dtype).cuda(), w.cuda(), b.cuda(), r.to(dtype).cuda()).float().cpu()
        outs.append(got)
    want = F.linear(q(x), q(w), b) + q(r)
    bad = (outs[0]-want).abs() > 0.1
    print(M,N,K,"bad elems", int(bad.sum()), "of", bad.numel(), "same across reps", bool((outs[0]==outs[1]).all()), bool((outs[1]==outs[2]).all()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print(" rows:", rows[:40].tolist(), "... n", len(rows)); print(" row%256 hist:", torch.bincount(rows%256, minlength=256).nonzero().flatten().tolist()[:64])
        print(" cols:", cols[:40].tolist(), "n", len(cols))
        print(" tiles(row//256) sample:", torch.unique(rows//256)[:40].tolist())
        ij = bad.nonzero()[:6]
        lin = F.linear(q(x), q(w), b)
        for (a, c) in ij.tolist():
            print("  bad at", a, c, "got", outs[0][a, c].item(), "want", want[a, c].item(), "lin", lin[a, c].item(), "res", q(r)[a, c].item(), "got-lin", outs[0][a,c].item()-lin[a,c].item(),
                  "res16rows", [round(q(r)[a + d, c].item(), 3) for d in (-48,-32,-16, 16, 32, 48) if 0 <= a + d < M])
        i=rows[0].item(); print(" row%16 hist", torch.bincount(rows%16, minlength=16).tolist()); print(" got", outs[0][i, cols[:8]].tolist(), "want", want[i, cols[:8]].tolist(), "res", q(r)[i, cols[:8]].tolist())
for a in [(71689,1152,128),(65536,1280,1280)]: run(*a)
