#!/usr/bin/env python3
"""Golden fixture for the DiT path (config 5): runs the REFERENCE's ``diffsim/diffsim_dit.py`` on the
reference's own vendored model ``DiT/modelsdit.py`` and ``DiT/diffusion`` (build container only).  timm is
not installed: ``timm.models.vision_transformer.{PatchEmbed, Attention, Mlp}`` are stand-ins restating
timm 1.0.12 semantics; diffusers' DDIMScheduler.add_noise and AutoencoderKL are stand-ins as in make_golden.py.
The reference hard-codes fp16 (``model.half()``, ``t_freq.to(torch.float16)``), so this fixture carries fp16
rounding: it pins STRUCTURE (block order, adaLN modulate, label/timestep conditioning, timestep_map quirk,
hook placement), and the tests compare against it with an fp16-sized tolerance.

Writes g9_dit_tiny.npz: scores for a few (layer, step, similarity) cases, q/k/v of image B for one case,
the latents and noise of the run, and the tiny model's state dict is the seeded synthetic one (seed 0).
"""
import io
import os
import sys
import types
import contextlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG                     # noqa: E402
import make_golden_xl as MX                  # noqa: E402
from oracle import cpu_ref as R             # noqa: E402
from diffsim_amd import config as C         # noqa: E402
from diffsim_amd import synth as S          # noqa: E402
from tests._fakes import FakeVAE            # noqa: E402


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim, bias=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, patch_size, stride=patch_size, bias=bias)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, **kw):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm, self.k_norm = nn.Identity(), nn.Identity()
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, Cc = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        x = F.scaled_dot_product_attention(self.q_norm(q), self.k_norm(k), v)
        return self.proj(x.transpose(1, 2).reshape(B, N, Cc))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _DDIM:
    def __init__(self):
        self.alphas_cumprod = R.alphas_cumprod()

    def add_noise(self, x, noise, timesteps):
        ac = self.alphas_cumprod.to(dtype=x.dtype)
        t = timesteps.reshape(-1).long()
        a, b = ac[t] ** 0.5, (1 - ac[t]) ** 0.5
        while a.ndim < x.ndim:
            a, b = a.unsqueeze(-1), b.unsqueeze(-1)
        return a * x + b * noise


def main():
    MX.install_xl_stubs()
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.PatchEmbed, vt.Attention, vt.Mlp = PatchEmbed, Attention, Mlp
    sys.modules["timm"] = types.ModuleType("timm")
    sys.modules["timm.models"] = types.ModuleType("timm.models")
    sys.modules["timm.models.vision_transformer"] = vt
    tv = types.ModuleType("torchvision.datasets.utils")
    tv.download_url = None
    sys.modules["torchvision.datasets"] = types.ModuleType("torchvision.datasets")
    sys.modules["torchvision.datasets.utils"] = tv

    import diffsim.diffsim_dit as ref_dit
    from DiT.modelsdit import DiT

    cfg = C.DIT_TINY
    sd = S.make_state_dict(cfg, seed=0)
    model = DiT(input_size=cfg.input_size, patch_size=cfg.patch_size, in_channels=cfg.in_channels,
                hidden_size=cfg.hidden_size, depth=cfg.depth, num_heads=cfg.num_heads, num_classes=cfg.num_classes)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all(k.startswith("final_layer") for k in missing.missing_keys), missing.missing_keys
    model.half().eval()

    dd = ref_dit.diffsim_DiT.__new__(ref_dit.diffsim_DiT)
    dd.model, dd.device, dd.scheduler = model, "cpu", _DDIM()
    dd.vae = FakeVAE()
    dd.vae.float = lambda: dd.vae
    img_a, img_b = os.path.join(HERE, "g1_img_c.png"), os.path.join(HERE, "g1_img_d.png")
    out = {}
    cases = [(2, 600, "cosine"), (2, 600, "mse"), (0, 750, "cosine"), (1, 900, "cosine")]
    for ci, (layer, step, sim) in enumerate(cases):
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            s = dd.diffsim_score(img_a, img_b, 128, "p", "none", [layer], step, sim, 2334)
        out[f"score_{ci}"] = np.asarray(s.float().numpy(), dtype=np.float32).reshape(-1)
        out[f"case_{ci}"] = np.array([str(layer), str(step), sim])
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        dd.diffsim_score(img_a, img_b, 128, "p", "none", [2], 600, "cosine", 2334)
    qb, kb, vb = model.blocks[2].attn.stores
    out["qB"], out["kB"], out["vB"] = (t.float().contiguous().numpy() for t in (qb, kb, vb))
    gen = ref_dit.get_generator(2334, "cpu")
    tA = ref_dit.process_image(ref_dit.load_image(img_a), 128)
    tB = ref_dit.process_image(ref_dit.load_image(img_b), 128)
    lA, lB = dd.prepare_image_latents(tA, gen), dd.prepare_image_latents(tB, gen)
    nA = torch.randn(lA.shape, generator=gen, dtype=torch.float16)          # randn_tensor(dtype=latents.dtype)
    nB = torch.randn(lB.shape, generator=gen, dtype=torch.float16)
    out["latA"], out["latB"] = lA.float().numpy(), lB.float().numpy()
    out["noiseA"], out["noiseB"] = nA.float().numpy(), nB.float().numpy()
    np.savez_compressed(os.path.join(HERE, "g9_dit_tiny.npz"), **out)
    print({k: v for k, v in out.items() if k.startswith("score")})


if __name__ == "__main__":
    main()
