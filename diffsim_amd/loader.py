"""Checkpoint loading: diffusers-layout model directories -> the engine's scorers.

The reference loads its pipelines with ``DiffSimPipeline.from_pretrained(<dir>, torch_dtype=torch.float16)``
(``/root/reference/diffsim/diffsim.py:80-82``, ``diffsim_xl.py:49``; the drivers hard-code NAS paths,
``cute_main.py:25-31``).  Here the same directory layout is read directly:

    <model_path>/unet/{config.json, diffusion_pytorch_model[.fp16].safetensors}
    <model_path>/vae/{config.json, diffusion_pytorch_model[.fp16].safetensors}
    <model_path>/text_encoder/{config.json, model[.fp16].safetensors}        (+ text_encoder_2 for SDXL)
    <model_path>/tokenizer/{vocab.json, merges.txt}                          (+ tokenizer_2 for SDXL)

Only file parsing happens here (safetensors -> CPU tensors); the scorers upload and repack the weights.  The tokenizer
is the one piece that needs the third-party vocabulary files: it is built lazily, on the first prompt, with
``transformers.CLIPTokenizer`` from the checkpoint's own ``tokenizer/`` folder.
"""
from __future__ import annotations

import dataclasses
import glob
import json
import os
from typing import Callable, Dict, Optional

import torch

from . import text as T
from .config import SD15, SDXL, VAE_SD15, DIT_XL2, DiTConfig, UNetConfig, VAEConfig


_WEIGHT_NAMES = ("diffusion_pytorch_model.safetensors", "model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                 "model.fp16.safetensors")


def _find_weights(folder: str):
    """The safetensors file(s) of a diffusers / transformers component folder: a single file (fp32 name first, then the
    fp16 variant), or every shard a ``*.safetensors.index.json`` lists (sharded checkpoints: ...-00001-of-0000N.safetensors)."""
    for name in _WEIGHT_NAMES:
        p = os.path.join(folder, name)
        if os.path.exists(p):
            return [p]
    # sharded checkpoints: probe the index names in the single-file order (fp32 first, then the fp16 variant) -- a plain sorted
    # glob would pick "...fp16.safetensors.index.json" before "...safetensors.index.json"; an index of another name is taken
    # only when it is the only one
    idx = [p for p in (os.path.join(folder, n + ".index.json") for n in _WEIGHT_NAMES) if os.path.exists(p)]
    if not idx:
        idx = sorted(glob.glob(os.path.join(folder, "*.safetensors.index.json")))
        if len(idx) > 1:
            raise FileNotFoundError(f"{folder} holds several *.safetensors.index.json files of unknown names: "
                                    f"{[os.path.basename(c) for c in idx]}")
    if idx:
        with open(idx[0]) as f:
            shards = sorted(set(json.load(f)["weight_map"].values()))
        paths = [os.path.join(folder, s) for s in shards]
        missing = [p for p in paths if not os.path.exists(p)]
        if missing:
            raise FileNotFoundError(f"{idx[0]} lists shards that are not there: {missing}")
        return paths
    cands = sorted(glob.glob(os.path.join(folder, "*.safetensors")))
    if not cands:
        raise FileNotFoundError(f"no .safetensors file under {folder}")
    if len(cands) > 1:
        raise FileNotFoundError(f"{folder} holds several .safetensors files and no *.safetensors.index.json saying which belong "
                                f"together: {[os.path.basename(c) for c in cands]}")
    return cands


def load_state_dict(folder: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    sd: Dict[str, torch.Tensor] = {}
    for p in _find_weights(folder):
        part = load_file(p, device="cpu")
        dup = set(part) & set(sd)
        if dup:
            raise ValueError(f"{p} repeats keys of an earlier shard: {sorted(dup)[:3]} ...")
        sd.update(part)
    return sd


def _json(folder: str) -> dict:
    p = os.path.join(folder, "config.json")
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        return json.load(f)


def unet_config_from_json(j: dict, default: UNetConfig) -> UNetConfig:
    """diffusers ``unet/config.json`` -> UNetConfig (only the fields the path depends on; absent = default)."""
    if not j:
        return default
    kw = {}
    for k in ("in_channels", "out_channels", "layers_per_block", "cross_attention_dim", "norm_num_groups", "norm_eps",
              "sample_size", "use_linear_projection", "addition_time_embed_dim"):
        if j.get(k) is not None:
            kw[k] = j[k]
    for k in ("block_out_channels", "down_block_types", "up_block_types"):
        if k in j:
            kw[k] = tuple(j[k])
    n = len(kw.get("block_out_channels", default.block_out_channels))
    ch = kw.get("block_out_channels", default.block_out_channels)
    ahd = j.get("attention_head_dim", None)
    if isinstance(ahd, (list, tuple)):          # SDXL: [5, 10, 20] = heads per level (diffusers' historical misnomer)
        kw["heads_per_level"] = tuple(int(v) for v in ahd)
    elif ahd is not None:
        kw["num_attention_heads"] = int(ahd)    # SD1.5: 8 = number of heads
    tl = j.get("transformer_layers_per_block", 1)
    if isinstance(tl, (list, tuple)):
        kw["depth_per_level"] = tuple(int(v) for v in tl)
    else:
        kw["transformer_layers_per_block"] = int(tl)
    if j.get("addition_embed_type") == "text_time":
        kw["addition_embed"] = True
        kw["sdxl_tap"] = True
        if j.get("projection_class_embeddings_input_dim"):
            kw["pooled_dim"] = int(j["projection_class_embeddings_input_dim"]) - 6 * int(kw.get("addition_time_embed_dim", 256))
    cfg = dataclasses.replace(default, **kw)
    assert len(cfg.down_block_types) == n and len(cfg.up_block_types) == n and len(ch) == n
    return cfg


def vae_config_from_json(j: dict, default: VAEConfig = VAE_SD15) -> VAEConfig:
    if not j:
        return default
    kw = {k: j[k] for k in ("in_channels", "latent_channels", "layers_per_block", "norm_num_groups", "scaling_factor") if k in j}
    if "block_out_channels" in j:
        kw["block_out_channels"] = tuple(j["block_out_channels"])
    return dataclasses.replace(default, **kw)


def clip_config_from_json(j: dict, default: T.CLIPTextConfig) -> T.CLIPTextConfig:
    if not j:
        return default
    m = {"vocab_size": "vocab_size", "hidden_size": "hidden_size", "intermediate_size": "intermediate_size",
         "num_hidden_layers": "num_layers", "num_attention_heads": "num_heads", "max_position_embeddings": "max_positions",
         "hidden_act": "act", "layer_norm_eps": "eps", "eos_token_id": "eos_token_id"}
    kw = {dst: j[src] for src, dst in m.items() if src in j}
    if default.projection_dim and "projection_dim" in j:
        kw["projection_dim"] = j["projection_dim"]
    return dataclasses.replace(default, **kw)


class LazyTokenizer:
    """``tokenize(str) -> LongTensor(1, max_length)`` from a checkpoint's tokenizer folder, built on first use
    (padding="max_length", truncation=True: StableDiffusionPipeline.encode_prompt's settings)."""

    def __init__(self, folder: str, max_length: int = 77):
        self.folder, self.max_length, self._tok = folder, max_length, None

    def __call__(self, prompt: str) -> torch.Tensor:
        if self._tok is None:
            if not os.path.exists(os.path.join(self.folder, "vocab.json")):
                raise FileNotFoundError(f"CLIP tokenizer files (vocab.json, merges.txt) not found under {self.folder}")
            from transformers import CLIPTokenizer
            self._tok = CLIPTokenizer.from_pretrained(self.folder)
        return self._tok(prompt, padding="max_length", max_length=self.max_length, truncation=True,
                         return_tensors="pt").input_ids


def _torch_dtype(name: str) -> torch.dtype:
    """Compute dtype of the kernels: bf16 (the headline MFMA mode), fp16 (the type the reference's drivers construct their
    pipelines in; same MFMA rate) or fp32 (parity mode).  The reference's fp16 *pipeline arithmetic* outside the U-Net
    (generator draws, add_noise) is --noise_dtype fp16."""
    try:
        return {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[name]
    except KeyError:
        raise ValueError(f"--dtype {name!r}: the engine computes in bf16, fp16 or fp32") from None


def load_diffsim(model_path: str, dtype: str = "bf16", device: str = "cuda", noise_dtype=torch.float32, **kw):
    """SD1.5 directory -> :class:`diffsim_amd.diffsim.DiffSim` with the HIP VAE encoder and the CLIP text encoder
    plugged in (what ``DiffSim(torch.float16, device)`` is in the reference, diffsim/diffsim.py:79-90)."""
    from .diffsim import DiffSim
    from .engine import VAEEncoder
    td = _torch_dtype(dtype)
    dev = "cuda:0" if device == "cuda" else device
    ucfg = unet_config_from_json(_json(os.path.join(model_path, "unet")), SD15)
    vcfg = vae_config_from_json(_json(os.path.join(model_path, "vae")))
    unet_sd = load_state_dict(os.path.join(model_path, "unet"))
    vae_sd = load_state_dict(os.path.join(model_path, "vae"))
    te_dir = os.path.join(model_path, "text_encoder")
    tcfg = clip_config_from_json(_json(te_dir), T.CLIP_L)
    te_sd = load_state_dict(te_dir)
    if tcfg.hidden_size != ucfg.cross_attention_dim:
        raise ValueError(f"text encoder width {tcfg.hidden_size} != unet cross_attention_dim {ucfg.cross_attention_dim}")
    ucfg = dataclasses.replace(ucfg, ctx_len=tcfg.max_positions)       # context length = the text encoder's positions (77)
    vae = VAEEncoder(vcfg, vae_sd, td, dev)                    # first GPU touch: raises DsimError without a GPU
    enc = T.CLIPTextEncoder(tcfg, te_sd, device=dev, dtype=torch.float32)
    encode = T.make_encode_prompt(enc, LazyTokenizer(os.path.join(model_path, "tokenizer"), tcfg.max_positions))
    return DiffSim(torch_dtype=td, device=dev, unet_config=ucfg, state_dict=unet_sd, vae=vae, encode_prompt=encode,
                   noise_dtype=noise_dtype, **kw)


def load_diffsim_xl(model_path: str, dtype: str = "bf16", device: str = "cuda", noise_dtype=torch.float32):
    """SDXL-base directory -> :class:`diffsim_amd.diffsim_xl.diffsim_xl` (diffsim/diffsim_xl.py:48-56)."""
    from .diffsim_xl import diffsim_xl
    from .engine import VAEEncoder
    td = _torch_dtype(dtype)
    dev = "cuda:0" if device == "cuda" else device
    ucfg = unet_config_from_json(_json(os.path.join(model_path, "unet")), SDXL)
    vcfg = vae_config_from_json(_json(os.path.join(model_path, "vae")), dataclasses.replace(VAE_SD15, scaling_factor=0.13025))
    unet_sd = load_state_dict(os.path.join(model_path, "unet"))
    vae = VAEEncoder(vcfg, load_state_dict(os.path.join(model_path, "vae")), torch.float32, dev)   # the reference runs the SDXL VAE in fp32
    e1 = T.CLIPTextEncoder(clip_config_from_json(_json(os.path.join(model_path, "text_encoder")), T.CLIP_L),
                           load_state_dict(os.path.join(model_path, "text_encoder")), device=dev)
    e2 = T.CLIPTextEncoder(clip_config_from_json(_json(os.path.join(model_path, "text_encoder_2")), T.OPENCLIP_BIGG),
                           load_state_dict(os.path.join(model_path, "text_encoder_2")), device=dev)
    encode = T.make_encode_prompt_xl(e1, e2, LazyTokenizer(os.path.join(model_path, "tokenizer")),
                                     LazyTokenizer(os.path.join(model_path, "tokenizer_2")))
    return diffsim_xl(td, dev, unet_config=ucfg, state_dict=unet_sd, vae=vae, encode_prompt=encode, noise_dtype=noise_dtype)


def load_diffsim_dit(model_path: str, img_size: int, target_step: int, dtype: str = "bf16", device: str = "cuda",
                     fp8_attention: bool = False):
    """``<model_path>/dit/*.safetensors`` (DiT/modelsdit.py keys) + ``<model_path>/vae`` -> diffsim_DiT
    (diffsim/diffsim_dit.py:30-61)."""
    from .diffsim_dit import diffsim_DiT
    from .engine import VAEEncoder
    td = _torch_dtype(dtype)
    dev = "cuda:0" if device == "cuda" else device
    j = _json(os.path.join(model_path, "dit"))
    cfg = dataclasses.replace(DIT_XL2, **{k: j[k] for k in dataclasses.asdict(DIT_XL2) if k in j})
    cfg = dataclasses.replace(cfg, input_size=img_size // 8)
    sd = load_state_dict(os.path.join(model_path, "dit"))
    vae = VAEEncoder(vae_config_from_json(_json(os.path.join(model_path, "vae"))), load_state_dict(os.path.join(model_path, "vae")),
                     torch.float32, dev)
    return diffsim_DiT(img_size, target_step, dev, dit_config=cfg, state_dict=sd, vae=vae, torch_dtype=td,
                       fp8_attention=fp8_attention)
