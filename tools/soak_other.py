"""Development probe: repeat-determinism of the VAE encoder and the DiT path (see soak_determinism.py)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.engine import VAEEncoder
from diffsim_amd.diffsim_dit import diffsim_DiT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), torch.bfloat16, "cuda")
x = torch.rand(8, 3, 512, 512, generator=torch.Generator().manual_seed(0)).mul(2).sub(1).cuda()
m0 = vae.moments(x).clone()
bad = sum(int(not torch.equal(vae.moments(x), m0)) for _ in range(N))
print(f"VAE: {N} repeats, {bad} mismatching")
cfg = C.DIT_XL2
keys = [k for k in C.dit_param_shapes(cfg) if not (k.startswith("blocks.") and int(k.split(".")[1]) > 13)]
for fp8 in (False, True):
    sc = diffsim_DiT(256, 600, "cuda", dit_config=cfg, state_dict=S.make_state_dict(cfg, seed=0, keys=keys), torch_dtype=torch.bfloat16,
                     fp8_attention=fp8)
    g = torch.Generator().manual_seed(5)
    zA, zB = torch.randn(32, 4, 32, 32, generator=g).cuda(), torch.randn(32, 4, 32, 32, generator=g).cuda()
    n = [t.cuda() for t in S.draw_pair_noise(2334, (1, 4, 32, 32))]
    s0 = sc.score_latent_pairs(zA, zB, n[2], n[3], 13, 600, "cosine", batch_pairs=32).clone()
    bad = sum(int(not torch.equal(sc.score_latent_pairs(zA, zB, n[2], n[3], 13, 600, "cosine", batch_pairs=32), s0)) for _ in range(N))
    print(f"DiT fp8={fp8}: {N} repeats, {bad} mismatching")
