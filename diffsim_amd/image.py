"""Host-side image preprocessing, mirroring the reference exactly.

``process_image``: /root/reference/diffsim/diffsim.py:27-41 (RGB -> Lanczos resize to a square ->
/255 -> (x-0.5)/0.5 -> NCHW fp32).  ``load_image``: diffusers.utils.load_image as the reference
uses it at diffsim/diffsim.py:103-104 (PIL open, EXIF transpose, RGB).
"""
from __future__ import annotations

import os

import numpy as np
import torch
from PIL import Image, ImageOps


def load_image(path_or_image):
    if isinstance(path_or_image, Image.Image):
        im = path_or_image
    else:
        im = Image.open(path_or_image)
    im = ImageOps.exif_transpose(im)
    return im.convert("RGB")


def resize_u8(image_, img_size: int = 512) -> torch.Tensor:
    """The host half of process_image: RGB -> Lanczos resize -> uint8 [1][H][W][3].  The arithmetic half (/255,
    (x - 0.5) / 0.5, NCHW, the fp16 cast) runs on the device (engine.image_preprocess), bit-identically."""
    image_ = image_.convert("RGB")
    image_ = image_.resize((img_size, img_size), resample=Image.Resampling.LANCZOS)
    return torch.from_numpy(np.array(image_)[None, :])


def process_image(image_, img_size: int = 512) -> torch.Tensor:
    image_ = image_.convert("RGB")
    image_ = image_.resize((img_size, img_size), resample=Image.Resampling.LANCZOS)
    arr = np.array(image_)[None, :].astype(np.float32) / 255.0
    arr = (arr - 0.5) / 0.5                      # VAE pixel range [-1, 1]
    return torch.from_numpy(arr.transpose(0, 3, 1, 2).copy())


def host_threads(world: int = 0) -> int:
    """Image-decode threads per rank.  PIL releases the GIL in decode and resize, so the threads scale with cores; the
    ranks of one node share the host (world = 0: read LOCAL_WORLD_SIZE / WORLD_SIZE), so that an 8-GPU node's eight
    ranks do not each start a full-width pool."""
    if world <= 0:
        world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    return max(4, min(64, n // max(1, world)))
