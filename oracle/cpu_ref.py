"""CPU oracle for the DiffSim scoring hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product package ``diffsim_amd`` never imports it and
fails loudly when its HIP extension is missing.

What it is: a plain PyTorch fp32 *CPU* restatement of the sub-graph the reference
executes for one DiffSim pair score:

  * one-step pipeline: PNDM timestep table + ``add_noise`` + CFG duplication
        reference: diffsim/diffsim_pipeline.py:140-221
  * SD1.5 ``UNet2DConditionModel`` forward up to (or past) the tapped attention
    layer; block control flow follows the only in-repo statement of it:
        diffsim/hacked_modules.py:17-136   (BasicTransformerBlock)
        diffsim/hacked_modules.py:261-434  (Transformer2DModel)
        diffsim/hacked_modules.py:438-535  (CrossAttnUpBlock2D)
        diffsim/hacked_modules.py:537-620  (CrossAttnDownBlock2D)
        diffsim/hacked_modules.py:622-688  (UNetMidBlock2DCrossAttn)
  * Q/K/V tap: diffsim/diffsim.py:43-56 -> diffsim/hacked_attn.py:61-77
  * score tail (4 SDPA + 2 cosine, or mse): diffsim/diffsim.py:177-197

PARITY PINNING.  The reference ships no tests, golden vectors or fixtures
 and the leaf arithmetic lives in un-vendored
``diffusers==0.29.2`` / ``torch==2.3.0`` (requirements.txt:2,18), absent from
/root/reference and from this image.  The leaves (ResnetBlock2D, GroupNorm/LayerNorm
eps, GEGLU, samplers, timestep embedding, PNDM table) are restated from the
published semantics of those versions.  The parts of the
path whose arithmetic IS in the reference's own files are pinned against outputs
of the reference itself, run in the build container by ``tests/golden/make_golden.py``
(fixtures under ``tests/golden/``): ``process_image``, generator draw order,
``hacked_AttnProcessor2_0`` q/k/v, the ``DiffSim.diffsim`` orchestration + score
tail, and the ``hacked_*_forward`` block control flow driven with this file's
modules.  The diffusers-internal leaves remain **parity unpinned**.

Module/parameter names follow the diffusers state-dict keys so a real SD1.5
``unet/diffusion_pytorch_model.safetensors`` loads with ``strict=True``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    """Subset of diffusers' ``unet/config.json`` that the path depends on."""

    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    # SD1.5's "attention_head_dim": 8 is really the number of heads (diffusers naming quirk)
    num_attention_heads: int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64
    transformer_layers_per_block: int = 1
    ctx_len: int = 77
    # ---- SDXL deltas; None/False = SD1.5 behaviour ---------------
    heads_per_level: Optional[Tuple[int, ...]] = None          # (5, 10, 20): head_dim 64
    depth_per_level: Optional[Tuple[int, ...]] = None          # transformer_layers_per_block (1, 2, 10)
    use_linear_projection: bool = False
    addition_embed: bool = False                                # addition_embed_type == "text_time"
    addition_time_embed_dim: int = 256
    pooled_dim: int = 1280                                      # text_embeds width (2816 = 1280 + 6*256)
    sdxl_tap: bool = False                                      # tap addressing of diffsim_xl.py:88-107

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def heads(self, level: int) -> int:
        return self.heads_per_level[level] if self.heads_per_level else self.num_attention_heads

    def depth(self, level: int) -> int:
        return self.depth_per_level[level] if self.depth_per_level else self.transformer_layers_per_block


SD15 = UNetConfig()
# small stand-in with the same topology (4 levels, 3 attention-bearing down/up blocks,
# straddling concat groups, 8x8 -> 1x1 ... kept >= 2x2 at the bottom) for fast tests
TINY = UNetConfig(block_out_channels=(64, 128, 256, 256), num_attention_heads=4,
                  cross_attention_dim=128, sample_size=16, ctx_len=13)
SDXL = UNetConfig(block_out_channels=(320, 640, 1280),
                  down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                  up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                  cross_attention_dim=2048, sample_size=128, heads_per_level=(5, 10, 20), depth_per_level=(1, 2, 10),
                  use_linear_projection=True, addition_embed=True, sdxl_tap=True)
# same topology (no attention at level 0, depths 1/2/3, linear projections, text_time embedding), small widths
SDXL_TINY = UNetConfig(block_out_channels=(64, 128, 256),
                       down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                       up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                       cross_attention_dim=128, sample_size=16, ctx_len=13, heads_per_level=(1, 2, 4),
                       depth_per_level=(1, 2, 3), use_linear_projection=True, addition_embed=True,
                       addition_time_embed_dim=32, pooled_dim=64, sdxl_tap=True)


# ----------------------------------------------------------------------------------------------
# scheduler facts (diffusers PNDMScheduler with SD1.5's scheduler_config)
# ----------------------------------------------------------------------------------------------
def alphas_cumprod(num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                   beta_end: float = 0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def pndm_timesteps(num_inference_steps: int = 1000, num_train_timesteps: int = 1000,
                   steps_offset: int = 1) -> np.ndarray:
    """``PNDMScheduler.set_timesteps`` with ``skip_prk_steps=True`` -> N+1 entries.

    reference call site: diffsim/diffsim_pipeline.py:153-157 (``timesteps[sample_timestep]``).
    """
    step_ratio = num_train_timesteps // num_inference_steps
    _t = (np.arange(0, num_inference_steps) * step_ratio).round() + steps_offset
    plms = np.concatenate([_t[:-1], _t[-2:-1], _t[-1:]])[::-1].copy()
    return plms.astype(np.int64)


def timestep_from_index(target_step: int) -> int:
    return int(pndm_timesteps()[target_step])


def add_noise(x0: torch.Tensor, noise: torch.Tensor, t: int) -> torch.Tensor:
    """``scheduler.add_noise``: sqrt(abar_t) x0 + sqrt(1-abar_t) eps (diffsim_pipeline.py:177-183)."""
    ac = alphas_cumprod()
    a = ac[t] ** 0.5
    b = (1.0 - ac[t]) ** 0.5
    return a * x0 + b * noise


def add_noise_f16(x0: torch.Tensor, noise: torch.Tensor, t: int) -> torch.Tensor:
    """PNDMScheduler.add_noise as the reference's fp16 pipeline runs it (diffsim_pipeline.py:177-183 on fp16
    latents): alphas_cumprod is cast to the sample dtype FIRST, the square roots, both products and the sum are
    fp16 operations.  Returns the fp16 result upcast (the oracle U-Net is fp32).  Pinned by fixture g10."""
    ac = alphas_cumprod().to(torch.float16)
    a, b = ac[t] ** 0.5, (1 - ac[t]) ** 0.5
    return (a * x0.to(torch.float16) + b * noise.to(torch.float16)).float()


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


# ----------------------------------------------------------------------------------------------
# leaves
# ----------------------------------------------------------------------------------------------
class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, temb_dim: int, groups: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x, output_size=None):
        # [EXT] diffusers 0.29.2 Upsample2D.forward: nearest x2, or nearest to an explicit size when the U-Net passes
        # upsample_size (latent sides that are not a multiple of 2**num_upsamplers, e.g. --image_size 224 -> 28 -> 14 -> 7 -> 4)
        if output_size is None:
            return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return self.conv(F.interpolate(x, size=output_size, mode="nearest"))


class Attention(nn.Module):
    """diffusers ``Attention`` with ``AttnProcessor2_0`` semantics (hacked_attn.py:38-101)."""

    def __init__(self, query_dim: int, heads: int, cross_dim: Optional[int] = None):
        super().__init__()
        self.heads = heads
        kv_dim = cross_dim if cross_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(kv_dim, query_dim, bias=False)
        self.to_v = nn.Linear(kv_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Dropout(0.0)])
        # attributes the reference's hacked processor reads (hacked_attn.py:39-99)
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = None

    def set_processor(self, processor):
        self.processor = processor

    def qkv(self, hidden_states, encoder_hidden_states=None):
        """(B,N,C) -> three (B,H,N,D) views, exactly hacked_attn.py:61-77."""
        b = hidden_states.shape[0]
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = self.to_q(hidden_states)
        k = self.to_k(ctx)
        v = self.to_v(ctx)
        d = k.shape[-1] // self.heads
        q = q.view(b, -1, self.heads, d).transpose(1, 2)
        k = k.view(b, -1, self.heads, d).transpose(1, 2)
        v = v.view(b, -1, self.heads, d).transpose(1, 2)
        return q, k, v

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        if self.processor is not None:  # used by the golden-capture cross-check only
            return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                  attention_mask=attention_mask, **kw)
        b = hidden_states.shape[0]
        q, k, v = self.qkv(hidden_states, encoder_hidden_states)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, q.shape[1] * q.shape[3])
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """norm1->attn1->+res ; norm2->attn2(ctx)->+res ; norm3->ff->+res (hacked_modules.py:39-132)."""

    def __init__(self, dim: int, heads: int, cross_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)
        # attributes read by hacked_BasicTransformerBlock_forward
        self.norm_type = "layer_norm"
        self.pos_embed = None
        self.only_cross_attention = False
        self._chunk_size = None
        self._chunk_dim = 0

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None,
                class_labels=None, added_cond_kwargs=None):
        h = hidden_states
        h = h + self.attn1(self.norm1(h))
        h = h + self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states)
        h = h + self.ff(self.norm3(h))
        return h


class Transformer2DModel(nn.Module):
    """continuous-input Transformer2D: GN(eps 1e-6) -> proj_in -> blocks -> proj_out -> +res."""

    def __init__(self, c: int, heads: int, cross_dim: int, groups: int, depth: int = 1,
                 use_linear_projection: bool = False):
        super().__init__()
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.proj_in = nn.Linear(c, c) if use_linear_projection else nn.Conv2d(c, c, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(c, heads, cross_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(c, c) if use_linear_projection else nn.Conv2d(c, c, 1)
        # attributes read by hacked_Transformer2DModel_forward (hacked_modules.py:336-429)
        self.is_input_continuous = True
        self.is_input_vectorized = False
        self.is_input_patches = False
        self.gradient_checkpointing = False

    def _operate_on_continuous_inputs(self, hidden_states):
        b, c, h, w = hidden_states.shape
        hidden_states = self.norm(hidden_states)
        if not self.use_linear_projection:
            hidden_states = self.proj_in(hidden_states)
            inner = hidden_states.shape[1]
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        else:
            inner = c
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, inner)
            hidden_states = self.proj_in(hidden_states)
        return hidden_states, inner

    def _get_output_for_continuous_inputs(self, hidden_states, residual, batch_size, height, width,
                                          inner_dim):
        if not self.use_linear_projection:
            hidden_states = hidden_states.reshape(batch_size, height, width, inner_dim)
            hidden_states = hidden_states.permute(0, 3, 1, 2).contiguous()
            hidden_states = self.proj_out(hidden_states)
        else:
            hidden_states = self.proj_out(hidden_states)
            hidden_states = hidden_states.reshape(batch_size, height, width, inner_dim)
            hidden_states = hidden_states.permute(0, 3, 1, 2).contiguous()
        return hidden_states + residual

    def forward(self, hidden_states, encoder_hidden_states=None, return_dict=False, **kw):
        b, _, h, w = hidden_states.shape
        residual = hidden_states
        x, inner = self._operate_on_continuous_inputs(hidden_states)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states)
        out = self._get_output_for_continuous_inputs(x, residual, b, h, w, inner)
        return (out,)


# ----------------------------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------------------------
class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, cin, cout, cfg: UNetConfig, add_downsample: bool, level: int = 0):
        super().__init__()
        n = cfg.layers_per_block
        self.resnets = nn.ModuleList([
            ResnetBlock2D(cin if i == 0 else cout, cout, cfg.time_embed_dim, cfg.norm_num_groups,
                          cfg.norm_eps) for i in range(n)])
        self.attentions = nn.ModuleList([
            Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim,
                               cfg.norm_num_groups, cfg.depth(level), cfg.use_linear_projection)
            for _ in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None
        self.has_cross_attention = True

    def forward(self, h, temb, ctx):
        outs = ()
        for res, attn in zip(self.resnets, self.attentions):
            h = res(h, temb)
            h = attn(h, encoder_hidden_states=ctx)[0]
            outs = outs + (h,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d(h)
            outs = outs + (h,)
        return h, outs


class DownBlock2D(nn.Module):
    def __init__(self, cin, cout, cfg: UNetConfig, add_downsample: bool, level: int = 0):
        super().__init__()
        n = cfg.layers_per_block
        self.resnets = nn.ModuleList([
            ResnetBlock2D(cin if i == 0 else cout, cout, cfg.time_embed_dim, cfg.norm_num_groups,
                          cfg.norm_eps) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None
        self.has_cross_attention = False

    def forward(self, h, temb, ctx=None):
        outs = ()
        for res in self.resnets:
            h = res(h, temb)
            outs = outs + (h,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d(h)
            outs = outs + (h,)
        return h, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, c, cfg: UNetConfig):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)
            for _ in range(2)])
        self.attentions = nn.ModuleList([
            Transformer2DModel(c, cfg.heads(len(cfg.block_out_channels) - 1), cfg.cross_attention_dim,
                               cfg.norm_num_groups, cfg.depth(len(cfg.block_out_channels) - 1),
                               cfg.use_linear_projection)])

    def forward(self, h, temb, ctx):
        h = self.resnets[0](h, temb)
        for attn, res in zip(self.attentions, self.resnets[1:]):
            h = attn(h, encoder_hidden_states=ctx)[0]
            h = res(h, temb)
        return h


def _up_resnets(cin, cout, prev, n, cfg):
    mods = []
    for i in range(n):
        skip = cin if i == n - 1 else cout
        rin = prev if i == 0 else cout
        mods.append(ResnetBlock2D(rin + skip, cout, cfg.time_embed_dim, cfg.norm_num_groups,
                                  cfg.norm_eps))
    return nn.ModuleList(mods)


class UpBlock2D(nn.Module):
    def __init__(self, cin, cout, prev, cfg: UNetConfig, add_upsample: bool, level: int = 0):
        super().__init__()
        self.resnets = _up_resnets(cin, cout, prev, cfg.layers_per_block + 1, cfg)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None
        self.has_cross_attention = False

    def forward(self, h, skips, temb, ctx=None, upsample_size=None):
        for res in self.resnets:
            s = skips[-1]
            skips = skips[:-1]
            h = res(torch.cat([h, s], dim=1), temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u(h, upsample_size)
        return h


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, cin, cout, prev, cfg: UNetConfig, add_upsample: bool, level: int = 0):
        super().__init__()
        n = cfg.layers_per_block + 1
        self.resnets = _up_resnets(cin, cout, prev, n, cfg)
        self.attentions = nn.ModuleList([
            Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim,
                               cfg.norm_num_groups, cfg.depth(level), cfg.use_linear_projection)
            for _ in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None
        self.has_cross_attention = True
        # attributes read by hacked_CrossAttnUpBlock2D_forward (hacked_modules.py:450-535)
        self.gradient_checkpointing = False
        self.use_ipa = False

    def forward(self, h, skips, temb, ctx, upsample_size=None):
        for res, attn in zip(self.resnets, self.attentions):
            s = skips[-1]
            skips = skips[:-1]
            h = res(torch.cat([h, s], dim=1), temb)
            h = attn(h, encoder_hidden_states=ctx)[0]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u(h, upsample_size)              # hacked_modules.py:531-533
        return h


class _TapReached(Exception):
    pass


class UNet2DConditionModel(nn.Module):
    """SD1.5-topology conditional U-Net."""

    def __init__(self, cfg: UNetConfig = SD15):
        super().__init__()
        self.cfg = cfg
        ch = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], cfg.time_embed_dim)
        if cfg.addition_embed:      # SDXL "text_time": pooled text embeds + 6 sinusoidal time ids
            self.add_embedding = TimestepEmbedding(cfg.pooled_dim + 6 * cfg.addition_time_embed_dim, cfg.time_embed_dim)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, typ in enumerate(cfg.down_block_types):
            cin, out = out, ch[i]
            last = i == len(ch) - 1
            cls = CrossAttnDownBlock2D if typ == "CrossAttnDownBlock2D" else DownBlock2D
            self.down_blocks.append(cls(cin, out, cfg, add_downsample=not last, level=i))
        self.mid_block = UNetMidBlock2DCrossAttn(ch[-1], cfg)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        out = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            last = i == len(ch) - 1
            cls = CrossAttnUpBlock2D if typ == "CrossAttnUpBlock2D" else UpBlock2D
            self.up_blocks.append(cls(cin, out, prev, cfg, add_upsample=not last, level=len(ch) - 1 - i))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    # -- tap selection: diffsim/diffsim.py:122-145 -------------------------------------------
    def tap_module(self, target_block: str, target_layer) -> Attention:
        if self.cfg.sdxl_tap:       # diffsim/diffsim_xl.py:88-107: [block, attention, transformer_block]
            tl = list(target_layer)
            if target_block == "down_blocks":
                return self.down_blocks[1:][tl[0]].attentions[tl[1]].transformer_blocks[tl[2]].attn1
            if target_block == "mid_blocks":
                return self.mid_block.attentions[tl[0]].transformer_blocks[tl[1]].attn1
            if target_block == "up_blocks":
                return self.up_blocks[:-1][tl[0]].attentions[tl[1]].transformer_blocks[tl[2]].attn1
            raise ValueError(target_block)
        if target_block == "down_blocks":
            blk = self.down_blocks[:-1][target_layer]
        elif target_block == "mid_blocks":
            blk = self.mid_block
        elif target_block == "up_blocks":
            blk = self.up_blocks[1:][target_layer]
        else:
            raise ValueError(target_block)
        return blk.attentions[-1].transformer_blocks[-1].attn1

    def forward(self, sample, t, ctx, stats: Optional[dict] = None, added_cond_kwargs: Optional[dict] = None):
        """Full forward to ``conv_out`` (what diffsim_pipeline.py:213-221 executes)."""
        b = sample.shape[0]
        wd = self.conv_in.weight.dtype        # float32; float64 when the oracle is evaluated in double (tests/probe_sdxl_f64.py)
        tt = torch.full((b,), float(t), dtype=torch.float32)
        temb = self.time_embedding(timestep_embedding(tt, self.cfg.block_out_channels[0]).to(wd))
        if self.cfg.addition_embed:
            te, ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
            tid = timestep_embedding(ids.flatten().float(), self.cfg.addition_time_embed_dim).reshape(b, -1)
            temb = temb + self.add_embedding(torch.cat([te.to(wd), tid.to(wd)], dim=-1))
        h = self.conv_in(sample)
        skips = (h,)
        _rec(stats, "conv_in", h)
        for i, blk in enumerate(self.down_blocks):
            h, outs = blk(h, temb, ctx)
            skips = skips + outs
            _rec(stats, f"down{i}", h)
        h = self.mid_block(h, temb, ctx)
        _rec(stats, "mid", h)
        # [EXT] UNet2DConditionModel.forward: a latent side that is not a multiple of 2**num_upsamplers makes every
        # non-final up block upsample to the size of the next skip instead of by 2 (forward_upsample_size)
        up_factor = 2 ** (len(self.up_blocks) - 1)
        forward_upsample_size = any(d % up_factor for d in sample.shape[-2:])
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            s, skips = skips[-n:], skips[:-n]
            size = tuple(skips[-1].shape[2:]) if (forward_upsample_size and i != len(self.up_blocks) - 1) else None
            h = blk(h, s, temb, ctx, upsample_size=size)
            _rec(stats, f"up{i}", h)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return h

    @torch.no_grad()
    def qkv_at_tap(self, sample, t, ctx, target_block="up_blocks", target_layer=0,
                   full: bool = False, stats: Optional[dict] = None, added_cond_kwargs: Optional[dict] = None):
        """Run the forward and return q,k,v (B,H,N,D) of the tapped attn1 (diffsim.py:43-56).

        ``full=False`` stops right after the tap (no numeric effect on q/k/v: nothing after the
        tap feeds it); ``full=True`` runs to ``conv_out`` like the reference does.
        """
        mod = self.tap_module(target_block, target_layer)
        store = {}

        def pre_hook(m, inp):
            store["qkv"] = m.qkv(inp[0])
            if stats is not None:
                _rec(stats, "tap_in", inp[0])
            if not full:
                raise _TapReached()

        hd = mod.register_forward_pre_hook(pre_hook)
        try:
            self.forward(sample, t, ctx, stats, added_cond_kwargs)
        except _TapReached:
            pass
        finally:
            hd.remove()
        return store["qkv"]


def _rec(stats, name, x):
    if stats is not None:
        xd = x.detach().double()
        stats[name] = [float(xd.mean()), float(xd.abs().mean()), float(xd.flatten()[::97].sum())]


# ----------------------------------------------------------------------------------------------
# score tail: diffsim/diffsim.py:177-197
# ----------------------------------------------------------------------------------------------
def pair_score(qa, ka, va, qb, kb, vb, similarity: str = "cosine") -> torch.Tensor:
    a_on_b = F.scaled_dot_product_attention(qa, kb, vb, dropout_p=0.0, is_causal=False)
    b_on_a = F.scaled_dot_product_attention(qb, ka, va, dropout_p=0.0, is_causal=False)
    self_a = F.scaled_dot_product_attention(qa, ka, va, dropout_p=0.0, is_causal=False)
    self_b = F.scaled_dot_product_attention(qb, kb, vb, dropout_p=0.0, is_causal=False)
    if similarity == "cosine":
        s1 = F.cosine_similarity(a_on_b.reshape(-1).unsqueeze(0), self_a.reshape(-1).unsqueeze(0))
        s2 = F.cosine_similarity(b_on_a.reshape(-1).unsqueeze(0), self_b.reshape(-1).unsqueeze(0))
    else:
        s1 = F.mse_loss(a_on_b, self_a)
        s2 = F.mse_loss(b_on_a, self_b)
    return (s1 + s2) / 2


# ----------------------------------------------------------------------------------------------
# one pair, latents-in: diffsim/diffsim.py:147-197 + diffsim_pipeline.py:140-221
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def features(unet: UNet2DConditionModel, z0, noise, ctx, target_step=600, target_block="up_blocks",
             target_layer=0, full=False, stats=None, fp16_pipeline=False):
    """z0, noise: (1,4,h,w); ctx: (2,L,Dc) = [uncond, cond] -> q,k,v each (2,H,N,D)."""
    t = timestep_from_index(target_step)
    xt = add_noise_f16(z0, noise, t) if fp16_pipeline else add_noise(z0, noise, t)
    xin = torch.cat([xt] * 2)          # CFG duplicate (diffsim_pipeline.py:208); PNDM scale = id
    return unet.qkv_at_tap(xin, t, ctx, target_block, target_layer, full=full, stats=stats)


@torch.no_grad()
def diffsim_latents(unet, zA, zB, nA, nB, ctx, target_step=600, target_block="up_blocks",
                    target_layer=0, similarity="cosine", full=False, fp16_pipeline=False) -> torch.Tensor:
    qa, ka, va = features(unet, zA, nA, ctx, target_step, target_block, target_layer, full, fp16_pipeline=fp16_pipeline)
    qb, kb, vb = features(unet, zB, nB, ctx, target_step, target_block, target_layer, full, fp16_pipeline=fp16_pipeline)
    return pair_score(qa, ka, va, qb, kb, vb, similarity)


def draw_pair_noise(seed: int, shape: Sequence[int]):
    """Reference draw order on ONE generator: vaeA, vaeB, noiseA, noiseB
    (diffsim/diffsim.py:109-113 then diffsim_pipeline.py:174-176 twice)."""
    g = torch.Generator("cpu").manual_seed(int(seed))
    return [torch.randn(tuple(shape), generator=g, dtype=torch.float32) for _ in range(4)]


def build_unet(cfg: UNetConfig, state_dict: dict) -> UNet2DConditionModel:
    m = UNet2DConditionModel(cfg)
    m.load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=True)
    return m.eval()


# ----------------------------------------------------------------------------------------------
# VAE encoder (reference call site diffsim/diffsim.py:92-96:
# ``pipe.vae.encode(image).latent_dist.sample(generator) * scaling_factor``).  diffusers'
# AutoencoderKL is not vendored: restated from SURVEY.md Appendix A item 11 -- PARITY UNPINNED.
# ----------------------------------------------------------------------------------------------
@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


VAE_SD15 = VAEConfig()
VAE_TINY = VAEConfig(block_out_channels=(64, 128, 256, 256))


class VAEResnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if self.conv_shortcut is not None else x) + h


class VAEDownsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class VAEDownBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([VAEResnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([VAEDownsample(cout)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class VAEAttention(nn.Module):
    """single-head spatial self-attention with GroupNorm and a residual connection"""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)
        return o + x


class VAEMidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([VAEResnet(c, c, groups), VAEResnet(c, c, groups)])
        self.attentions = nn.ModuleList([VAEAttention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class VAEEncoderNet(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.down_blocks.append(VAEDownBlock(prev, c, cfg.layers_per_block, cfg.norm_num_groups, i != len(ch) - 1))
            prev = c
        self.mid_block = VAEMidBlock(ch[-1], cfg.norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLEncoder(nn.Module):
    """encode(x) -> moments (mean, logvar); sample = mean + exp(0.5*clamp(logvar,-30,20)) * randn(generator)"""

    def __init__(self, cfg: VAEConfig = VAE_SD15):
        super().__init__()
        self.cfg = cfg
        self.encoder = VAEEncoderNet(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    @torch.no_grad()
    def moments(self, x):
        return self.quant_conv(self.encoder(x.float()))

    @torch.no_grad()
    def sample(self, x, generator=None):
        mean, logvar = self.moments(x).chunk(2, dim=1)
        std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
        return mean + std * torch.randn(mean.shape, generator=generator, dtype=mean.dtype)


# ----------------------------------------------------------------------------------------------
# SDXL one-step pipeline arithmetic (diffsim/diffsim_xl_pipeline.py:190-323): EulerDiscreteScheduler
# with SDXL's scheduler_config (scaled_linear betas, timestep_spacing "leading", steps_offset 1).
# diffusers is not vendored: restated from SURVEY.md Appendix A item 14 -- PARITY UNPINNED.
# ----------------------------------------------------------------------------------------------
def euler_tables(num_inference_steps: int = 1000, num_train_timesteps: int = 1000, steps_offset: int = 1):
    """-> (timesteps float32 [N], sigmas float32 [N+1], init_noise_sigma)."""
    ac = alphas_cumprod().numpy().astype(np.float64)
    sig_all = ((1 - ac) / ac) ** 0.5
    ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32) + steps_offset
    sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    init = float((sig.max() ** 2 + 1) ** 0.5)      # "leading" spacing
    return ts, sig, init


def sdxl_inputs(z0: torch.Tensor, noise: torch.Tensor, target_step: int):
    """Reference quirk reproduced (diffsim_xl_pipeline.py:204-225, 309): the CLEAN latents are
    multiplied by init_noise_sigma, then sigma*noise is added, then the sum is divided by
    sqrt(sigma^2+1).  Returns (x_in, t)."""
    ts, sig, init = euler_tables()
    t, s = float(ts[target_step]), float(sig[target_step])
    x = z0 * init + noise * s
    return x / ((s * s + 1) ** 0.5), t


def sdxl_inputs_f16(z0: torch.Tensor, noise: torch.Tensor, target_step: int):
    """The same chain on fp16 tensors, as the reference's fp16 pipeline runs it: prepare_latents multiplies the fp16
    latents by the python float init_noise_sigma; add_noise casts the sigma table to fp16; scale_model_input divides
    by a 0-dim fp32 tensor (the fp16 sample keeps its dtype).  Pinned by fixture g10."""
    ts, sig, init = euler_tables()
    t = float(ts[target_step])
    sg = torch.tensor(float(sig[target_step]), dtype=torch.float32)
    x = z0.to(torch.float16) * init
    x = x + noise.to(torch.float16) * sg.to(torch.float16)
    x = x / ((sg ** 2 + 1) ** 0.5)
    return x.float(), t


def sdxl_time_ids(cfg: UNetConfig) -> torch.Tensor:
    side = float(cfg.sample_size * 8)      # height/width default to sample_size * vae_scale_factor
    return torch.tensor([[side, side, 0.0, 0.0, side, side]], dtype=torch.float32)


@torch.no_grad()
def features_xl(unet: UNet2DConditionModel, z0, noise, ctx, pooled, target_step, target_block, target_layer, full=False,
                fp16_pipeline=False):
    """z0, noise (1,4,h,w) at ANY latent side (time_ids stay the model's native size); ctx (2,L,Dc) = [neg, pos];
    pooled (2,P) = [neg, pos] -> q,k,v (2,H,N,D)."""
    x, t = sdxl_inputs_f16(z0, noise, target_step) if fp16_pipeline else sdxl_inputs(z0, noise, target_step)
    xin = torch.cat([x] * 2)
    added = {"text_embeds": pooled, "time_ids": sdxl_time_ids(unet.cfg).repeat(2, 1)}
    return unet.qkv_at_tap(xin, t, ctx, target_block, target_layer, full=full, added_cond_kwargs=added)


@torch.no_grad()
def diffsim_xl_latents(unet, zA, zB, nA, nB, ctx, pooled, target_step, target_block, target_layer,
                       similarity="cosine", full=False, fp16_pipeline=False):
    a = features_xl(unet, zA, nA, ctx, pooled, target_step, target_block, target_layer, full, fp16_pipeline)
    b = features_xl(unet, zB, nB, ctx, pooled, target_step, target_block, target_layer, full, fp16_pipeline)
    return pair_score(*a, *b, similarity)


# ----------------------------------------------------------------------------------------------
# DiT-XL/2 path: diffsim/diffsim_dit.py:74-142 on DiT/modelsdit.py.
# The model file is in the reference (restated here from DiT/modelsdit.py:20-21,28-66,103-124,147-250,
# 278-325 with timm's Attention / Mlp / PatchEmbed semantics, which are NOT vendored); the timestep
# respacing follows DiT/diffusion/respace.py:12-129 and is pinned by fixture G7.
# ----------------------------------------------------------------------------------------------
@dataclass
class DiTConfig:
    input_size: int = 32            # latent side (256 px / 8)
    patch_size: int = 2
    in_channels: int = 4
    hidden_size: int = 1152
    depth: int = 28
    num_heads: int = 16
    mlp_ratio: int = 4
    num_classes: int = 1000
    freq_dim: int = 256


DIT_XL2 = DiTConfig()
DIT_TINY = DiTConfig(input_size=16, hidden_size=128, depth=3, num_heads=4)


def dit_pos_embed(dim: int, grid: int) -> torch.Tensor:
    """get_2d_sincos_pos_embed (DiT/modelsdit.py:278-325): (1, grid*grid, dim) f32."""
    def one_d(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gh, gh), axis=0).reshape(2, 1, grid, grid)      # w first
    emb = np.concatenate([one_d(dim // 2, g[0]), one_d(dim // 2, g[1])], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0)


def dit_timestep_map(target_step: int, num_timesteps: int = 1000):
    """SpacedDiffusion(space_timesteps(1000, str(target_step))).timestep_map (respace.py:12-88)."""
    n = int(target_step)
    stride = 1 if n <= 1 else (num_timesteps - 1) / (n - 1)
    return sorted({round(i * stride) for i in range(n)})


class _DiTAttention(nn.Module):          # timm Attention: fused qkv with bias, identity q/k norm
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads, self.head_dim = heads, dim // heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.q_norm, self.k_norm = nn.Identity(), nn.Identity()

    def split(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        return qkv.unbind(0)

    def forward(self, x):
        b, n, c = x.shape
        q, k, v = self.split(x)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.proj(o.transpose(1, 2).reshape(b, n, c))


class _DiTMlp(nn.Module):                # timm Mlp with tanh-GELU
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x), approximate="tanh"))


class _DiTBlock(nn.Module):
    def __init__(self, dim, heads, ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.attn = _DiTAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.mlp = _DiTMlp(dim, dim * ratio)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim))

    def forward(self, x, c):
        sm, cm, gm, sl, cl, gl = self.adaLN_modulation(c).chunk(6, dim=1)
        x = x + gm.unsqueeze(1) * self.attn(self.norm1(x) * (1 + cm.unsqueeze(1)) + sm.unsqueeze(1))
        x = x + gl.unsqueeze(1) * self.mlp(self.norm2(x) * (1 + cl.unsqueeze(1)) + sl.unsqueeze(1))
        return x


class _TEmb(nn.Module):
    def __init__(self, dim, fdim):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(fdim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.fdim = fdim

    def forward(self, t):
        half = self.fdim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        args = t[:, None].float() * freqs[None]
        return self.mlp(torch.cat([torch.cos(args), torch.sin(args)], dim=-1))


class DiTOracle(nn.Module):
    def __init__(self, cfg: DiTConfig = DIT_XL2):
        super().__init__()
        self.cfg = cfg
        d, p = cfg.hidden_size, cfg.patch_size
        self.x_embedder = nn.Module()
        self.x_embedder.proj = nn.Conv2d(cfg.in_channels, d, p, stride=p)
        self.t_embedder = _TEmb(d, cfg.freq_dim)
        self.y_embedder = nn.Module()
        self.y_embedder.embedding_table = nn.Embedding(cfg.num_classes + 1, d)
        g = cfg.input_size // p
        self.pos_embed = nn.Parameter(dit_pos_embed(d, g), requires_grad=False)
        self.blocks = nn.ModuleList([_DiTBlock(d, cfg.num_heads, cfg.mlp_ratio) for _ in range(cfg.depth)])

    @torch.no_grad()
    def qkv_at(self, x, t_model: int, y, layer: int):
        """x (1,C,H,W); y (2,) class ids -> q,k,v (2,H,N,D) = inputs of blocks[layer].attn (diffsim_dit.py:19-26)."""
        h = self.x_embedder.proj(x).flatten(2).transpose(1, 2) + self.pos_embed
        c = self.t_embedder(torch.tensor([float(t_model)])) + self.y_embedder.embedding_table(y)
        for i, blk in enumerate(self.blocks):
            if i == layer:
                sm, cm = blk.adaLN_modulation(c).chunk(6, dim=1)[:2]
                return blk.attn.split(blk.norm1(h) * (1 + cm.unsqueeze(1)) + sm.unsqueeze(1))
            h = blk(h, c)
        raise IndexError(layer)


@torch.no_grad()
def dit_features(model: DiTOracle, z0, noise, target_step: int, layer: int):
    """diffsim_dit.py:87-114: noise at t = target_step with the SD1.5 DDIM alphas, model conditioned on
    timestep_map[1000 - target_step], labels [1, num_classes] (class 1 + null)."""
    ac = alphas_cumprod()
    xt = ac[target_step] ** 0.5 * z0 + (1 - ac[target_step]) ** 0.5 * noise
    tm = dit_timestep_map(target_step)[1000 - target_step]
    y = torch.tensor([1, model.cfg.num_classes])
    return model.qkv_at(xt, tm, y, layer)


@torch.no_grad()
def diffsim_dit_latents(model, zA, zB, nA, nB, target_step, layer, similarity="cosine"):
    a = dit_features(model, zA, nA, target_step, layer)
    b = dit_features(model, zB, nB, target_step, layer)
    return pair_score(*a, *b, similarity)
