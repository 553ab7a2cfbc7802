"""Development probe: per-kernel-family time of one forward at a small batch (HIP events per launch)."""
import sys, collections
import torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = C.SD15
ds = DiffSim(torch.bfloat16, "cuda", state_dict=S.make_state_dict(cfg, seed=0))
ctx = S.make_context(cfg).cuda()
n = [t.cuda() for t in S.draw_pair_noise(2334, (1, 4, 64, 64))]
prs = [S.make_pair_latents(cfg, i) for i in range(bp)]
zA, zB = torch.cat([p[0] for p in prs]).cuda(), torch.cat([p[1] for p in prs]).cuda()
for _ in range(2):
    ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, batch_pairs=bp)
eng = ds.engine("up_blocks", 0)
eng.profile(True)
ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, batch_pairs=bp)
recs = eng.profile_records()
eng.profile(False)
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, fl, by, ms in recs:
    f = fam[name]; f[0] += 1; f[1] += fl; f[2] += ms
tot = sum(v[2] for v in fam.values())
print("total kernel ms", round(tot, 3))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:36s} n={v[0]:3d} ms={v[2]:7.3f}  {v[1]/max(v[2],1e-9)/1e9:8.1f} TF/s")
# the slowest individual launches
for name, fl, by, ms in sorted(recs, key=lambda r: -r[3])[:14]:
    print(f"   {name:36s} {ms*1e3:8.1f} us  {fl/1e9:8.2f} GF")
