"""Small-batch latency: eager launches vs hipGraph replay (DiffSim(use_graphs=True)).  Run on the GPU box."""
import time, torch, sys
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim
cfg = C.SD15
keys = None
sd = S.make_state_dict(cfg, seed=0)
ctx = S.make_context(cfg).cuda()
shp = (1, 4, 64, 64)
n = [t.cuda() for t in S.draw_pair_noise(2334, shp)]
for npairs in (1, 2, 4):
    prs = [S.make_pair_latents(cfg, i) for i in range(npairs)]
    zA, zB = torch.cat([p[0] for p in prs]).cuda(), torch.cat([p[1] for p in prs]).cuda()
    res = {}
    for g in (False, True):
        ds = DiffSim(torch.bfloat16, "cuda", state_dict=sd, use_graphs=g)
        for _ in range(3):
            s = ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, "up_blocks", 0, 600, "cosine", batch_pairs=npairs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            s = ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, "up_blocks", 0, 600, "cosine", batch_pairs=npairs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        res[g] = (dt * 1e3, s.cpu().tolist())
        del ds
    print(npairs, "eager %.3f ms  graph %.3f ms" % (res[False][0], res[True][0]), "equal", res[False][1] == res[True][1], flush=True)
