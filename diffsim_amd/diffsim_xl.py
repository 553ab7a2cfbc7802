"""DiffSim-XL scorer (SDXL U-Net) with the reference's entry points, backed by the MI355X engine.

Mirrors ``/root/reference/diffsim/diffsim_xl.py`` (``diffsim_xl.__init__`` :48-56, ``prepare_image_latents``
:58-63, ``diffsim_score`` :65-155) and what ``DiffSimXLPipeline.step`` does around the U-Net call
(``/root/reference/diffsim/diffsim_xl_pipeline.py:163-323``): prompt + pooled embeddings, Euler
index -> (t, sigma), latents * init_noise_sigma + sigma*noise, / sqrt(sigma^2+1), CFG duplication,
``added_cond_kwargs = {text_embeds, time_ids}``.  ``target_layer`` is the reference's 3-int address
``[block, attention, transformer_block]`` (2 ints for mid_blocks).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from . import scheduler as sched
from .config import SDXL, UNetConfig
from .diffsim import get_generator
from .engine import UNetEngine, pair_score
from .image import load_image, process_image


class diffsim_xl:
    def __init__(self, torch_dtype=torch.bfloat16, device="cuda", ip_adapter=False, *, unet_config: UNetConfig = SDXL,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, vae=None,
                 encode_prompt: Optional[Callable[[str], Tuple[torch.Tensor, torch.Tensor]]] = None,
                 noise_dtype=torch.float32):
        if ip_adapter:
            raise NotImplementedError("IP-Adapter mode is out of scope")
        if state_dict is None:
            raise ValueError("state_dict (diffusers-keyed SDXL U-Net weights) is required")
        self.dtype = torch_dtype
        self.device = torch.device("cuda:0" if device == "cuda" else device)
        self.ip_adapter = False
        self.cfg, self.state_dict, self.vae, self._encode_prompt = unet_config, state_dict, vae, encode_prompt
        # float32: the reference's arithmetic carried out in fp32 (draws included) -- the parity setting.
        # float16: the literal fp16 pipeline -- latents arrive fp16 (diffsim_xl.py:63), prepare_latents multiplies them
        # by init_noise_sigma in fp16, randn_tensor(dtype=latents.dtype) draws in fp16 (a different random stream),
        # add_noise and scale_model_input run in fp16 (diffsim_xl_pipeline.py:204-225, 309)
        if noise_dtype not in (torch.float32, torch.float16):
            raise ValueError("noise_dtype must be torch.float32 or torch.float16")
        self.noise_dtype = noise_dtype
        self._base: Optional[UNetEngine] = None
        self._engines: Dict[tuple, object] = {}

    def engine(self, target_block: str, target_layer):
        """The engine positioned at a tap; one packed weight copy serves every tap."""
        key = (target_block, tuple(int(v) for v in target_layer))
        if key not in self._engines:
            if self._base is None:
                self._base = UNetEngine(self.cfg, self.state_dict, self.dtype, target_block, list(key[1]), str(self.device))
            self._engines[key] = self._base.view(target_block, list(key[1]))
            self._engines[key].tokens
        return self._engines[key]

    def prepare_image_latents(self, image, generator=None):
        if self.vae is None:
            raise RuntimeError("no VAE plugged in: use score_latent_pairs")
        lat = self.vae.encode(image.to(dtype=torch.float32)).latent_dist.sample(generator=generator)
        lat = self.vae.config.scaling_factor * lat
        return lat.to(dtype=torch.float16)                         # diffsim_xl.py:63

    def time_ids(self) -> torch.Tensor:
        # original_size = target_size = (height, width) = unet.config.sample_size * vae_scale_factor -- the model's
        # NATIVE size (1024 for SDXL), whatever --image_size the latents were encoded at: step() is called without
        # height/width (diffsim_xl.py:109-125 -> diffsim_xl_pipeline.py:127-131, 231-246)
        side = float(self.cfg.sample_size * 8)
        return torch.tensor([[side, side, 0.0, 0.0, side, side]] * 2, dtype=torch.float32)

    @torch.no_grad()
    def features(self, latents, noise, ctx, pooled, target_block, target_layer, target_step):
        """latents / noise (n,4,s,s): any latent side s (img_size // 8), not only cfg.sample_size."""
        eng = self.engine(target_block, target_layer)
        t, a, b = sched.sdxl_step_coefficients(int(target_step))
        if self.noise_dtype == torch.float16:       # the fp16 text encoders' outputs
            ctx, pooled = ctx.to(torch.float16).float(), pooled.to(torch.float16).float()
        eng.set_conditioning(t, pooled, self.time_ids())
        ctx = ctx.to(self.device, torch.float32).contiguous()
        if self.noise_dtype == torch.float16:
            ts, sig, init = sched.euler_tables()
            sg = torch.tensor(float(sig[int(target_step)]), dtype=torch.float32)
            dev = self.device
            x = latents.to(dev, torch.float16) * init                          # prepare_latents: fp16 * python float
            x = x + noise.to(dev, torch.float16) * sg.to(torch.float16).to(dev)  # add_noise: sigmas cast to the sample dtype
            x = x / ((sg ** 2 + 1) ** 0.5).to(dev)                              # scale_model_input: fp16 / 0-dim fp32 -> fp16
            x = x.float().contiguous()
            return eng.qkv(x, torch.zeros_like(x), 1.0, 0.0, ctx)
        return eng.qkv(latents.to(self.device, torch.float32).contiguous(), noise.to(self.device, torch.float32).contiguous(),
                       a, b, ctx)

    @torch.no_grad()
    def score_latent_pairs(self, latA, latB, noiseA, noiseB, ctx, pooled, target_block, target_layer, target_step,
                           similarity="cosine", batch_pairs: int = 8) -> torch.Tensor:
        n = latA.shape[0]
        eng = self.engine(target_block, target_layer)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        shp = latA.shape[1:]
        batch_pairs = max(1, min(batch_pairs, eng.max_images() // 2))      # every activation must stay < 2 GiB
        for i0 in range(0, n, batch_pairs):
            i1 = min(n, i0 + batch_pairs)
            m = i1 - i0
            lat = torch.stack([latA[i0:i1], latB[i0:i1]], dim=1).reshape(2 * m, *shp).float()
            nA = noiseA[i0:i1] if (noiseA.shape[0] == n and n > 1) else noiseA.expand(m, *shp)
            nB = noiseB[i0:i1] if (noiseB.shape[0] == n and n > 1) else noiseB.expand(m, *shp)
            nz = torch.stack([nA, nB], dim=1).reshape(2 * m, *shp)
            q, k, v = self.features(lat, nz, ctx, pooled, target_block, target_layer, target_step)
            ia = torch.arange(0, 2 * m, 2, dtype=torch.int32, device=self.device)
            out[i0:i1] = pair_score(q, k, v, ia, ia + 1, eng.heads, similarity)
        return out

    @torch.no_grad()
    def diffsim_score(self, image_A, image_B, img_size, prompt, target_block, target_layer, target_step, similarity, seed):
        """Same contract as the reference's ``diffsim_xl.diffsim_score`` (diffsim/diffsim_xl.py:65-155)."""
        if self._encode_prompt is None:
            raise RuntimeError("no text encoder plugged in: pass encode_prompt=...")
        tensor_A, tensor_B = process_image(load_image(image_A), img_size), process_image(load_image(image_B), img_size)
        generator = get_generator(seed, "cpu")
        latentsA = self.prepare_image_latents(tensor_A, generator)
        latentsB = self.prepare_image_latents(tensor_B, generator)
        noiseA = torch.randn(latentsA.shape, generator=generator, dtype=self.noise_dtype).float()
        noiseB = torch.randn(latentsB.shape, generator=generator, dtype=self.noise_dtype).float()
        ctx, pooled = self._encode_prompt(prompt)
        return self.score_latent_pairs(latentsA.float(), latentsB.float(), noiseA, noiseB, ctx, pooled, target_block,
                                       target_layer, target_step, similarity)
