"""Development probe: files-in throughput of DiffSim.score_pairs (PNG decode + Lanczos resize on the host thread pool,
VAE + U-Net + tail on the GPU), synthetic weights, 64 random 600x500 PNG pairs."""
import os, sys, time, tempfile
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim
from diffsim_amd.engine import VAEEncoder

d = tempfile.mkdtemp(); rng = np.random.default_rng(0); paths = []
for i in range(32):
    p = os.path.join(d, f"im{i}.png"); Image.fromarray(rng.integers(0, 255, (500, 600, 3), dtype=np.uint8)).save(p); paths.append(p)
pairs = [(paths[i % 32], paths[(i * 7 + 3) % 32]) for i in range(64)]
vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), torch.bfloat16, "cuda")
ctx = S.make_context(C.SD15)
ds = DiffSim(torch.bfloat16, "cuda", state_dict=S.make_state_dict(C.SD15, seed=0), vae=vae, encode_prompt=lambda p: ctx)
ds.score_pairs(pairs[:16], 512, "a photo", "up_blocks", 0, 600, seed=2334, batch_pairs=16)
torch.cuda.synchronize(); t0 = time.perf_counter()
s = ds.score_pairs(pairs, 512, "a photo", "up_blocks", 0, 600, seed=2334, batch_pairs=16)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"files-in: {len(pairs)} pairs in {dt*1e3:.0f} ms -> {len(pairs)/dt:.1f} pairs/s  (host cores {os.cpu_count()})", s[:3].tolist())
