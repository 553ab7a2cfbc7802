"""Development probe: VAE encoder on synthetic 512 px images (for rocprofv3 --kernel-trace --stats)."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
from diffsim_amd import config as C, synth as S
from diffsim_amd.engine import VAEEncoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
vae = VAEEncoder(C.VAEConfig(), S.make_state_dict(C.VAEConfig(), seed=1), torch.bfloat16, "cuda")
x = torch.rand(n, 3, 512, 512, generator=torch.Generator().manual_seed(0)).mul(2).sub(1).cuda()
for _ in range(2):
    m = vae.moments(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    m = vae.moments(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"{n} images: {dt*1e3:.2f} ms  -> {n/dt:.1f} images/s, {n * 1116.66e9 / dt / 1e12:.1f} TFLOP/s", float(m.float().abs().mean()))
