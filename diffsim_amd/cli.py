"""Command-line flags of the reference drivers (/root/reference/argprocess.py:5-18), same names
and defaults, plus the handful the MI355X build adds (model path, dtype, batch, GPUs)."""
from __future__ import annotations

import argparse


def arg_parse(argv=None):
    p = argparse.ArgumentParser(description="DiffSim scoring (MI355X-native engine)")
    p.add_argument("--image_path", type=str, help="Path to image folder")
    p.add_argument("--original_path", type=str, default=None, help="Path to original images for ipref")
    p.add_argument("--out_path", type=str, help="Path to the output folder")
    p.add_argument("--image_size", type=int, default=512, help="(Resized) resolution of compared image")
    p.add_argument("--target_block", type=str, choices=["down_blocks", "mid_blocks", "up_blocks"], default="up_blocks")
    p.add_argument("--target_layer", type=int, default=2, nargs="+")
    p.add_argument("--target_step", type=int, default=100)
    p.add_argument("--metric", type=str, default="diffsim",
                   choices=["diffsim", "diffsim_xl", "clip_i", "clip_cross", "dino", "dinov1", "dino_cross", "cute",
                            "lpips", "gram", "diffeats", "clipfeats", "dinofeats", "ensemble", "dit"])
    p.add_argument("--similarity", type=str, choices=["cosine", "mse"], default="mse")
    p.add_argument("--prompt", type=str, default="High quality image")
    p.add_argument("--ip_adapter", action="store_true")
    p.add_argument("--use_mask", action="store_true")
    p.add_argument("--use_text_attn", action="store_true")
    p.add_argument("--seed", type=int, default=2333)
    # additions of this build
    p.add_argument("--model_path", type=str, default=None, help="diffusers-layout SD1.5 directory (unet/*.safetensors)")
    p.add_argument("--dtype", type=str, choices=["bf16", "fp32"], default="bf16")
    p.add_argument("--batch", type=int, default=16, help="pairs per U-Net batch")
    return p.parse_args(argv)
