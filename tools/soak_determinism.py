"""Development probe: the same step N times -- q/k/v and scores must be bit-identical every time (a race in the LDS ring /
persistent tile hand-off / attention staging / the small-batch kernel's counted-vmcnt ring would show up as a flipped bit).
    python tools/soak_determinism.py [N=200] [pairs=32] [bf16|fp16]      (pairs = 1 exercises the small-batch GEMM)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 32
DT = torch.float16 if len(sys.argv) > 3 and sys.argv[3] == "fp16" else torch.bfloat16
cfg = C.SD15
ds = DiffSim(DT, "cuda", state_dict=S.make_state_dict(cfg, seed=0))
ctx = S.make_context(cfg).cuda()
n = [t.cuda() for t in S.draw_pair_noise(2334, (1, 4, 64, 64))]
prs = [S.make_pair_latents(cfg, i) for i in range(P)]
zA, zB = torch.cat([p[0] for p in prs]).cuda(), torch.cat([p[1] for p in prs]).cuda()
lat = torch.stack([zA, zB], dim=1).reshape(2 * P, 4, 64, 64); nz = torch.stack([n[2].expand(P, -1, -1, -1), n[3].expand(P, -1, -1, -1)], dim=1).reshape(2 * P, 4, 64, 64)
q0, k0, v0 = (t.clone() for t in ds.features(lat, nz, ctx, "up_blocks", 0, 600))
s0 = ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, batch_pairs=P).clone()
bad = 0
for i in range(N):
    q, k, v = ds.features(lat, nz, ctx, "up_blocks", 0, 600)
    s = ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, batch_pairs=P)
    if not (torch.equal(q, q0) and torch.equal(k, k0) and torch.equal(v, v0) and torch.equal(s, s0)):
        bad += 1
print(f"{N} repeats of {P} pair(s) in {DT}, {bad} mismatching", "OK" if bad == 0 else "FAIL")
