"""U-Net topology description for the DiffSim scoring path.

Mirrors the fields of diffusers' ``unet/config.json`` that the reference's pipeline relies
on (reference: diffsim/diffsim.py:82 loads the SD1.5 pipeline; SURVEY.md Appendix A lists
the SD1.5 values).  Parameter names produced by :func:`unet_param_shapes` are the diffusers
state-dict keys, so a real ``diffusion_pytorch_model.safetensors`` can be fed unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    num_attention_heads: int = 8          # SD1.5 "attention_head_dim": 8 == number of heads
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64                  # latent side (image side / 8)
    transformer_layers_per_block: int = 1
    ctx_len: int = 77
    # ---- SDXL deltas; None/False = SD1.5 behaviour -----------------
    heads_per_level: Optional[Tuple[int, ...]] = None       # (5, 10, 20): head_dim 64
    depth_per_level: Optional[Tuple[int, ...]] = None       # transformer_layers_per_block (1, 2, 10)
    use_linear_projection: bool = False
    addition_embed: bool = False                             # addition_embed_type "text_time"
    addition_time_embed_dim: int = 256
    pooled_dim: int = 1280
    sdxl_tap: bool = False                                   # [block, attention, transformer_block] addressing

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def heads(self, level: int) -> int:
        return self.heads_per_level[level] if self.heads_per_level else self.num_attention_heads

    def depth(self, level: int) -> int:
        return self.depth_per_level[level] if self.depth_per_level else self.transformer_layers_per_block


SD15 = UNetConfig()
#: same channel plan as SD1.5 on an 8x8 latent (64 px image): every kernel shape family of the
#: real model with tiny M -- the GPU parity workhorse.
SD15_SMALL = UNetConfig(sample_size=8)
#: small-channel stand-in with identical topology (fast CPU tests)
TINY = UNetConfig(block_out_channels=(64, 128, 256, 256), num_attention_heads=4,
                  cross_attention_dim=128, sample_size=16, ctx_len=13)
#: SDXL base U-Net (BASELINE.json config 4): 3 levels, no attention at level 0, transformer depth 1/2/10
SDXL = UNetConfig(block_out_channels=(320, 640, 1280),
                  down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                  up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                  cross_attention_dim=2048, sample_size=128, heads_per_level=(5, 10, 20), depth_per_level=(1, 2, 10),
                  use_linear_projection=True, addition_embed=True, sdxl_tap=True)
SDXL_TINY = UNetConfig(block_out_channels=(64, 128, 256),
                       down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                       up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                       cross_attention_dim=128, sample_size=16, ctx_len=13, heads_per_level=(1, 2, 4),
                       depth_per_level=(1, 2, 3), use_linear_projection=True, addition_embed=True,
                       addition_time_embed_dim=32, pooled_dim=64, sdxl_tap=True)


def _resnet(p: str, cin: int, cout: int, temb: int, out: Dict[str, Tuple[int, ...]]):
    out[p + "norm1.weight"] = (cin,)
    out[p + "norm1.bias"] = (cin,)
    out[p + "conv1.weight"] = (cout, cin, 3, 3)
    out[p + "conv1.bias"] = (cout,)
    out[p + "time_emb_proj.weight"] = (cout, temb)
    out[p + "time_emb_proj.bias"] = (cout,)
    out[p + "norm2.weight"] = (cout,)
    out[p + "norm2.bias"] = (cout,)
    out[p + "conv2.weight"] = (cout, cout, 3, 3)
    out[p + "conv2.bias"] = (cout,)
    if cin != cout:
        out[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[p + "conv_shortcut.bias"] = (cout,)


def _transformer(p: str, c: int, cfg: UNetConfig, out: Dict[str, Tuple[int, ...]], level: int = 0):
    proj = (c, c) if cfg.use_linear_projection else (c, c, 1, 1)
    out[p + "norm.weight"] = (c,)
    out[p + "norm.bias"] = (c,)
    out[p + "proj_in.weight"] = proj
    out[p + "proj_in.bias"] = (c,)
    for j in range(cfg.depth(level)):
        q = f"{p}transformer_blocks.{j}."
        for n in ("norm1", "norm2", "norm3"):
            out[q + n + ".weight"] = (c,)
            out[q + n + ".bias"] = (c,)
        out[q + "attn1.to_q.weight"] = (c, c)
        out[q + "attn1.to_k.weight"] = (c, c)
        out[q + "attn1.to_v.weight"] = (c, c)
        out[q + "attn1.to_out.0.weight"] = (c, c)
        out[q + "attn1.to_out.0.bias"] = (c,)
        out[q + "attn2.to_q.weight"] = (c, c)
        out[q + "attn2.to_k.weight"] = (c, cfg.cross_attention_dim)
        out[q + "attn2.to_v.weight"] = (c, cfg.cross_attention_dim)
        out[q + "attn2.to_out.0.weight"] = (c, c)
        out[q + "attn2.to_out.0.bias"] = (c,)
        out[q + "ff.net.0.proj.weight"] = (8 * c, c)
        out[q + "ff.net.0.proj.bias"] = (8 * c,)
        out[q + "ff.net.2.weight"] = (c, 4 * c)
        out[q + "ff.net.2.bias"] = (c,)
    out[p + "proj_out.weight"] = proj
    out[p + "proj_out.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """All parameters of the full U-Net, diffusers keys, in forward order."""
    out: Dict[str, Tuple[int, ...]] = {}
    ch = cfg.block_out_channels
    temb = cfg.time_embed_dim
    out["conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3)
    out["conv_in.bias"] = (ch[0],)
    out["time_embedding.linear_1.weight"] = (temb, ch[0])
    out["time_embedding.linear_1.bias"] = (temb,)
    out["time_embedding.linear_2.weight"] = (temb, temb)
    out["time_embedding.linear_2.bias"] = (temb,)
    if cfg.addition_embed:
        ain = cfg.pooled_dim + 6 * cfg.addition_time_embed_dim
        out["add_embedding.linear_1.weight"] = (temb, ain)
        out["add_embedding.linear_1.bias"] = (temb,)
        out["add_embedding.linear_2.weight"] = (temb, temb)
        out["add_embedding.linear_2.bias"] = (temb,)
    prev = ch[0]
    for i, typ in enumerate(cfg.down_block_types):
        cin, prev = prev, ch[i]
        for j in range(cfg.layers_per_block):
            _resnet(f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else ch[i], ch[i], temb, out)
            if typ == "CrossAttnDownBlock2D":
                _transformer(f"down_blocks.{i}.attentions.{j}.", ch[i], cfg, out, i)
        if i != len(ch) - 1:
            out[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (ch[i], ch[i], 3, 3)
            out[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (ch[i],)
    c = ch[-1]
    _resnet("mid_block.resnets.0.", c, c, temb, out)
    _transformer("mid_block.attentions.0.", c, cfg, out, len(ch) - 1)
    _resnet("mid_block.resnets.1.", c, c, temb, out)
    rev = list(reversed(ch))
    o = rev[0]
    n = cfg.layers_per_block + 1
    for i, typ in enumerate(cfg.up_block_types):
        prv, o = o, rev[i]
        cin = rev[min(i + 1, len(ch) - 1)]
        for j in range(n):
            skip = cin if j == n - 1 else o
            rin = prv if j == 0 else o
            _resnet(f"up_blocks.{i}.resnets.{j}.", rin + skip, o, temb, out)
            if typ == "CrossAttnUpBlock2D":
                _transformer(f"up_blocks.{i}.attentions.{j}.", o, cfg, out, len(ch) - 1 - i)
        if i != len(ch) - 1:
            out[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (o, o, 3, 3)
            out[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (o,)
    out["conv_norm_out.weight"] = (ch[0],)
    out["conv_norm_out.bias"] = (ch[0],)
    out["conv_out.weight"] = (cfg.out_channels, ch[0], 3, 3)
    out["conv_out.bias"] = (cfg.out_channels,)
    return out


def skip_channel_plan(cfg: UNetConfig) -> List[int]:
    """Channels of the 12 skip tensors pushed on the down path."""
    ch = cfg.block_out_channels
    skips = [ch[0]]
    for i in range(len(ch)):
        skips += [ch[i]] * cfg.layers_per_block
        if i != len(ch) - 1:
            skips.append(ch[i])
    return skips


@dataclass(frozen=True)
class VAEConfig:
    """Encoder half of diffusers' AutoencoderKL config (SD1.5 ``vae/config.json``)."""
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


VAE_SD15 = VAEConfig()
VAE_TINY = VAEConfig(block_out_channels=(64, 128, 256, 256))


def _vae_resnet(p: str, cin: int, cout: int, out: Dict[str, Tuple[int, ...]]):
    out[p + "norm1.weight"] = (cin,)
    out[p + "norm1.bias"] = (cin,)
    out[p + "conv1.weight"] = (cout, cin, 3, 3)
    out[p + "conv1.bias"] = (cout,)
    out[p + "norm2.weight"] = (cout,)
    out[p + "norm2.bias"] = (cout,)
    out[p + "conv2.weight"] = (cout, cout, 3, 3)
    out[p + "conv2.bias"] = (cout,)
    if cin != cout:
        out[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[p + "conv_shortcut.bias"] = (cout,)


def vae_encoder_param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    """Encoder + quant_conv parameters of AutoencoderKL under their diffusers keys."""
    out: Dict[str, Tuple[int, ...]] = {}
    ch = cfg.block_out_channels
    out["encoder.conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3)
    out["encoder.conv_in.bias"] = (ch[0],)
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _vae_resnet(f"encoder.down_blocks.{i}.resnets.{j}.", prev if j == 0 else c, c, out)
        if i != len(ch) - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
        prev = c
    c = ch[-1]
    _vae_resnet("encoder.mid_block.resnets.0.", c, c, out)
    a = "encoder.mid_block.attentions.0."
    out[a + "group_norm.weight"] = (c,)
    out[a + "group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[a + n + ".weight"] = (c, c)
        out[a + n + ".bias"] = (c,)
    _vae_resnet("encoder.mid_block.resnets.1.", c, c, out)
    out["encoder.conv_norm_out.weight"] = (c,)
    out["encoder.conv_norm_out.bias"] = (c,)
    m = 2 * cfg.latent_channels
    out["encoder.conv_out.weight"] = (m, c, 3, 3)
    out["encoder.conv_out.bias"] = (m,)
    out["quant_conv.weight"] = (m, m, 1, 1)
    out["quant_conv.bias"] = (m,)
    return out


@dataclass(frozen=True)
class DiTConfig:
    """DiT-XL/2 as instantiated by the reference (diffsim/diffsim_dit.py:31-35, DiT/modelsdit.py:147-176)."""
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


def dit_param_shapes(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    """Parameters of DiT up to the last block (final_layer is never needed), reference key names."""
    d, p = cfg.hidden_size, cfg.patch_size
    t = (cfg.input_size // p) ** 2
    out: Dict[str, Tuple[int, ...]] = {}
    out["pos_embed"] = (1, t, d)
    out["x_embedder.proj.weight"] = (d, cfg.in_channels, p, p)
    out["x_embedder.proj.bias"] = (d,)
    out["t_embedder.mlp.0.weight"] = (d, cfg.freq_dim)
    out["t_embedder.mlp.0.bias"] = (d,)
    out["t_embedder.mlp.2.weight"] = (d, d)
    out["t_embedder.mlp.2.bias"] = (d,)
    out["y_embedder.embedding_table.weight"] = (cfg.num_classes + 1, d)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        out[b + "attn.qkv.weight"] = (3 * d, d)
        out[b + "attn.qkv.bias"] = (3 * d,)
        out[b + "attn.proj.weight"] = (d, d)
        out[b + "attn.proj.bias"] = (d,)
        out[b + "mlp.fc1.weight"] = (cfg.mlp_ratio * d, d)
        out[b + "mlp.fc1.bias"] = (cfg.mlp_ratio * d,)
        out[b + "mlp.fc2.weight"] = (d, cfg.mlp_ratio * d)
        out[b + "mlp.fc2.bias"] = (d,)
        out[b + "adaLN_modulation.1.weight"] = (6 * d, d)
        out[b + "adaLN_modulation.1.bias"] = (6 * d,)
    return out
