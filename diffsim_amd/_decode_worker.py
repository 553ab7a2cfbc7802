"""Image decode + Lanczos resize in worker PROCESSES (no torch, no GPU in here: only PIL and numpy).

A thread pool stops scaling at ~3 cores on this step: PIL's decoders are fed from a Python loop and EXIF handling, mode
conversion and the ndarray copy hold the GIL in between.  Worker processes scale with the cores; each returns the uint8
HWC pixels (the arithmetic half of process_image runs on the device, engine.image_preprocess)."""
import numpy as np
from PIL import Image, ImageOps


def decode_resize(path: str, img_size: int) -> np.ndarray:
    """diffusers.utils.load_image (open, EXIF transpose, RGB) + the resize of process_image
    (/root/reference/diffsim/diffsim.py:27-33, 103-104) -> uint8 [H][W][3]."""
    im = Image.open(path)
    im = ImageOps.exif_transpose(im)
    im = im.convert("RGB")
    im = im.resize((img_size, img_size), resample=Image.Resampling.LANCZOS)
    return np.asarray(im)


def decode_resize_many(paths, img_size: int) -> np.ndarray:
    """One task = a few images: fewer, larger messages between the processes."""
    return np.stack([decode_resize(p, img_size) for p in paths])


def serve() -> None:
    """Worker loop (``python -c "from diffsim_amd._decode_worker import serve; serve()"``): one request per line on stdin,
    ``<img_size>\\t<path>``; answer on stdout: an 8-byte little-endian length (0 = failure, followed by a length-prefixed
    utf-8 message) and the raw uint8 pixels.  Exits at EOF."""
    import struct
    import sys
    out = sys.stdout.buffer
    for line in sys.stdin.buffer:
        size, _, path = line.rstrip(b"\n").partition(b"\t")
        try:
            px = np.ascontiguousarray(decode_resize(path.decode(), int(size)))
            out.write(struct.pack("<q", px.nbytes))
            out.write(px.data)
        except Exception as e:                              # the parent raises with this text
            msg = f"{type(e).__name__}: {e}".encode()
            out.write(struct.pack("<q", 0) + struct.pack("<q", len(msg)) + msg)
        out.flush()
