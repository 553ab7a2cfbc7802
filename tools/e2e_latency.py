"""Development probe: wall time of one DiffSim.diffsim(path_A, path_B, ...) call (the reference's per-pair API) with the
VAE encoder and image preprocessing included; synthetic weights, random 600x500 PNGs."""
import os, sys, time, tempfile
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim
from diffsim_amd.engine import VAEEncoder

d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
paths = []
for i in range(2):
    p = os.path.join(d, f"im{i}.png")
    Image.fromarray(rng.integers(0, 255, (500, 600, 3), dtype=np.uint8)).save(p)
    paths.append(p)
vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), torch.bfloat16, "cuda")
ctx = S.make_context(C.SD15)
ds = DiffSim(torch.bfloat16, "cuda", state_dict=S.make_state_dict(C.SD15, seed=0), vae=vae, encode_prompt=lambda p: ctx)
for _ in range(3):
    s = ds.diffsim(paths[0], paths[1], 512, "a photo", "up_blocks", 0, 600, seed=2334)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
n = 10
for _ in range(n):
    s = ds.diffsim(paths[0], paths[1], 512, "a photo", "up_blocks", 0, 600, seed=2334)
    float(s)
dt = (time.perf_counter() - t0) / n
print(f"diffsim() per call: {dt*1e3:.1f} ms  score {float(s):.6f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    float(ds.diffsim(paths[0], paths[1], 512, "a photo", "up_blocks", 0, 600, seed=2334))
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(22); print(buf.getvalue()[-4200:])
