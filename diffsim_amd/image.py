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
        # after parallel.pin_to_gpu_numa the affinity mask already is one NUMA node's cores: divide it by the ranks on that
        # node, not by the whole local world a second time
        world = int(os.environ.get("DSIM_RANKS_ON_THESE_CPUS", os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    return max(4, min(64, n // max(1, world)))


class DecodePool:
    """Decode + resize of image files ahead of the GPU: ``submit(paths, size)`` -> futures whose results concatenate to the
    uint8 pixels [n][H][W][3] of `paths` (``gather``).

    procs > 0: that many worker PROCESSES (plain children running ``_decode_worker.serve`` over pipes: PIL and numpy only, no
    torch, no GPU, nothing inherited from a process that has initialised HIP) -- the form that scales with the host's
    cores.  procs = 0: threads of this process (no start-up cost; fine for a few images, GIL-bound at ~3 cores).
    Default: DSIM_DECODE_PROCS (a number, or "auto" = cores / ranks of the node), else threads."""

    def __init__(self, procs=None, threads=None):
        import queue
        from concurrent.futures import ThreadPoolExecutor
        if procs is None:
            e = os.environ.get("DSIM_DECODE_PROCS", "0")
            procs = host_threads() if e == "auto" else int(e)
        self.procs = int(procs)
        self._workers = None
        self._free = queue.Queue()
        self._ex = ThreadPoolExecutor(max_workers=self.procs if self.procs > 0 else (threads or host_threads()))

    def _start(self):
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        self._workers = [subprocess.Popen([sys.executable, "-c", "from diffsim_amd._decode_worker import serve; serve()"],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env) for _ in range(self.procs)]
        for w in self._workers:
            self._free.put(w)

    def _one(self, path: str, img_size: int) -> np.ndarray:
        if self.procs <= 0:
            from ._decode_worker import decode_resize
            return decode_resize(path, img_size)[None]
        import struct
        w = self._free.get()
        try:
            w.stdin.write(f"{int(img_size)}\t{path}\n".encode())
            w.stdin.flush()
            (n,) = struct.unpack("<q", w.stdout.read(8))
            if n == 0:
                (m,) = struct.unpack("<q", w.stdout.read(8))
                raise RuntimeError(f"decode worker failed on {path}: {w.stdout.read(m).decode()}")
            buf = w.stdout.read(n)
            return np.frombuffer(buf, dtype=np.uint8).reshape(1, img_size, img_size, 3)
        finally:
            self._free.put(w)

    def submit(self, paths, img_size: int):
        if self.procs > 0 and self._workers is None:
            self._start()
        return [self._ex.submit(self._one, p, img_size) for p in paths]

    @staticmethod
    def gather(futures) -> torch.Tensor:
        return torch.from_numpy(np.concatenate([f.result() for f in futures]))

    def shutdown(self):
        self._ex.shutdown(wait=False, cancel_futures=True)
        for w in self._workers or []:
            try:
                w.stdin.close()
                w.terminate()
            except Exception:
                pass
        self._workers = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
