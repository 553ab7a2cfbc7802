"""Development probe: where the wall time of one per-pair diffsim() call goes (explicit synchronisation per stage)."""
import os, sys, time, tempfile
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim, get_generator
from diffsim_amd.image import load_image, process_image
from diffsim_amd.engine import VAEEncoder

d = tempfile.mkdtemp(); rng = np.random.default_rng(0); paths = []
for i in range(2):
    p = os.path.join(d, f"im{i}.png"); Image.fromarray(rng.integers(0, 255, (500, 600, 3), dtype=np.uint8)).save(p); paths.append(p)
vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), torch.bfloat16, "cuda")
ctx = S.make_context(C.SD15)
ds = DiffSim(torch.bfloat16, "cuda", state_dict=S.make_state_dict(C.SD15, seed=0), vae=vae, encode_prompt=lambda p: ctx)
for _ in range(3):
    ds.diffsim(paths[0], paths[1], 512, "a photo", "up_blocks", 0, 600, seed=2334)
torch.cuda.synchronize()
acc = {}
def lap(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
N = 10
for _ in range(N):
    t = time.perf_counter()
    A, B = load_image(paths[0]), load_image(paths[1]); t = lap("load_image x2", t)
    tA, tB = process_image(A, 512), process_image(B, 512); t = lap("process_image x2", t)
    hA, hB = tA.to(dtype=torch.float16), tB.to(dtype=torch.float16); t = lap("to fp16 (cpu) x2", t)
    dA = hA.to("cuda").float(); dB = hB.to("cuda").float(); t = lap("H2D + float x2", t)
    mA = vae.moments(dA); t = lap("vae.moments A", t)
    mB = vae.moments(dB); t = lap("vae.moments B", t)
    g = get_generator(2334, "cpu")
    from diffsim_amd.engine import _LatentDist
    lA = 0.18215 * _LatentDist(mA).sample(g); lB = 0.18215 * _LatentDist(mB).sample(g); t = lap("sample x2 (cpu randn + H2D)", t)
    nA = torch.randn(lA.shape, generator=g); nB = torch.randn(lB.shape, generator=g); t = lap("noise randn cpu", t)
    s = ds.score_latent_pairs(lA.float(), lB.float(), nA, nB, "a photo", "up_blocks", 0, 600, "cosine"); t = lap("score_latent_pairs", t)
    v = float(s); t = lap("float(score)", t)
tot = sum(acc.values())
for k, v in acc.items(): print(f"{k:32s} {v / N * 1e3:7.2f} ms")
print(f"{'total':32s} {tot / N * 1e3:7.2f} ms")
