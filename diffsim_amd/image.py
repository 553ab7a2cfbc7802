"""Host-side image preprocessing, mirroring the reference exactly.

``process_image``: /root/reference/diffsim/diffsim.py:27-41 (RGB -> Lanczos resize to a square ->
/255 -> (x-0.5)/0.5 -> NCHW fp32).  ``load_image``: diffusers.utils.load_image as the reference
uses it at diffsim/diffsim.py:103-104 (PIL open, EXIF transpose, RGB).
"""
from __future__ import annotations

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


def process_image(image_, img_size: int = 512) -> torch.Tensor:
    image_ = image_.convert("RGB")
    image_ = image_.resize((img_size, img_size), resample=Image.Resampling.LANCZOS)
    arr = np.array(image_)[None, :].astype(np.float32) / 255.0
    arr = (arr - 0.5) / 0.5                      # VAE pixel range [-1, 1]
    return torch.from_numpy(arr.transpose(0, 3, 1, 2).copy())
