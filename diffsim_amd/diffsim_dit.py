"""DiffSim-DiT scorer with the reference's entry points, backed by the MI355X engine.

Mirrors ``/root/reference/diffsim/diffsim_dit.py``: ``diffsim_DiT.__init__`` :30-61 (model + VAE + DDIM scheduler),
``prepare_image_latents`` :54-59, ``add_noise`` :63-72, ``diffsim_score`` :74-142 (labels [1, 1000], hook on
``model.blocks[target_layer[0]].attn``, ``p_sample`` at t = 1000 - target_step, shared score tail).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import scheduler as sched
from .config import DIT_XL2, DiTConfig
from .diffsim import get_generator
from .engine import DiTEngine, pair_score
from .image import load_image, process_image


class diffsim_DiT:
    def __init__(self, img_size=256, target_step=600, device="cuda", ckpt=None, *, dit_config: DiTConfig = DIT_XL2,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, vae=None, torch_dtype=torch.bfloat16,
                 fp8_attention: bool = False):
        if state_dict is None:
            raise ValueError("state_dict (DiT weights under DiT/modelsdit.py keys) is required; no checkpoint is bundled")
        if img_size // 8 != dit_config.input_size:
            raise ValueError("img_size must be 8 * dit_config.input_size")
        self.cfg, self.state_dict, self.vae = dit_config, state_dict, vae
        self.dtype = torch_dtype
        self.device = torch.device("cuda:0" if device == "cuda" else device)
        self.fp8_attention = fp8_attention      # e4m3 MFMAs for QK^T and PV inside the DiT blocks (opt-in)
        self._engine: Optional[DiTEngine] = None

    def engine(self, layer: int) -> DiTEngine:
        """The engine with its tap at blocks[layer]: ONE packed weight copy, the tap is moved (dsim_dit_set_tap)."""
        if self._engine is None:
            self._engine = DiTEngine(self.cfg, self.state_dict, self.dtype, int(layer), str(self.device))
            if self.fp8_attention:
                self._engine.set_attention(True)
        self._engine.set_tap(int(layer))
        return self._engine

    def prepare_image_latents(self, image, generator=None):
        if self.vae is None:
            raise RuntimeError("no VAE plugged in: use score_latent_pairs")
        lat = self.vae.encode(image.to(dtype=torch.float32)).latent_dist.sample(generator=generator)
        return (self.vae.config.scaling_factor * lat).to(dtype=torch.float16)      # diffsim_dit.py:59

    @torch.no_grad()
    def features(self, latents, noise, target_layer: int, target_step: int):
        eng = self.engine(int(target_layer))
        eng.set_conditioning(sched.dit_model_timestep(int(target_step)), 1, self.cfg.num_classes)   # y = [1, null]
        sa, sb = sched.noise_coefficients(int(target_step))         # DDIM add_noise at t = target_step (SD1.5 betas)
        return eng.qkv(latents.to(self.device, torch.float32).contiguous(), noise.to(self.device, torch.float32).contiguous(),
                       sa, sb)

    @torch.no_grad()
    def score_latent_pairs(self, latA, latB, noiseA, noiseB, target_layer: int, target_step: int, similarity="cosine",
                           batch_pairs: int = 32) -> torch.Tensor:
        n = latA.shape[0]
        eng = self.engine(int(target_layer))
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        shp = latA.shape[1:]
        for i0 in range(0, n, batch_pairs):
            i1 = min(n, i0 + batch_pairs)
            m = i1 - i0
            lat = torch.stack([latA[i0:i1], latB[i0:i1]], dim=1).reshape(2 * m, *shp).float()
            nz = torch.stack([noiseA.expand(m, *shp), noiseB.expand(m, *shp)], dim=1).reshape(2 * m, *shp).float()
            q, k, v = self.features(lat, nz, target_layer, target_step)
            ia = torch.arange(0, 2 * m, 2, dtype=torch.int32, device=self.device)
            out[i0:i1] = pair_score(q, k, v, ia, ia + 1, eng.heads, similarity)
        return out

    @torch.no_grad()
    def diffsim_score(self, image_A, image_B, img_size, prompt, target_block, target_layer, target_step, similarity, seed):
        """Same contract as the reference's ``diffsim_DiT.diffsim_score`` (diffsim/diffsim_dit.py:74-142)."""
        layer = target_layer[0]
        tA, tB = process_image(load_image(image_A), img_size), process_image(load_image(image_B), img_size)
        generator = get_generator(seed, "cpu")
        latentsA = self.prepare_image_latents(tA, generator)
        latentsB = self.prepare_image_latents(tB, generator)
        # randn_tensor(dtype=latents.dtype): the reference draws the noise in fp16 (diffsim_dit.py:64-66)
        noiseA = torch.randn(latentsA.shape, generator=generator, dtype=latentsA.dtype)
        noiseB = torch.randn(latentsB.shape, generator=generator, dtype=latentsB.dtype)
        return self.score_latent_pairs(latentsA.float(), latentsB.float(), noiseA.float(), noiseB.float(), layer, target_step,
                                       similarity)
