"""Synthetic weights and inputs for the DiffSim scoring path.

No SD1.5 checkpoint or dataset exists offline, so benchmarks and parity tests use seeded
random weights of the real architecture (diffusers-keyed state dict) and image-like latents.
Everything is drawn with ``torch.Generator('cpu')`` so the GPU box and the build container
produce bit-identical tensors.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .config import DiTConfig, UNetConfig, VAEConfig, dit_param_shapes, unet_param_shapes, vae_encoder_param_shapes


def make_state_dict(cfg: UNetConfig, seed: int = 0, keys: Optional[Sequence[str]] = None
                    ) -> Dict[str, torch.Tensor]:
    """Seeded fp32 state dict.  Variance-preserving uniform init (U(+-sqrt(3/fan_in))) so
    activations neither vanish nor explode through ~60 layers; ``to_q``/``to_k`` carry gain
    1.4 so attention logits have std ~2 (non-degenerate softmax; SURVEY.md section 7 "hard parts");
    norm affine parameters are perturbed away from (1,0) so a dropped gamma/beta is caught.
    Each tensor has its own generator keyed by its position, so a subset is reproducible."""
    if isinstance(cfg, DiTConfig):
        shapes = dit_param_shapes(cfg)
    else:
        shapes = vae_encoder_param_shapes(cfg) if isinstance(cfg, VAEConfig) else unet_param_shapes(cfg)
    out: Dict[str, torch.Tensor] = {}
    for idx, (k, shp) in enumerate(shapes.items()):
        if keys is not None and k not in keys:
            continue
        g = torch.Generator("cpu").manual_seed(seed * 1000003 + idx)
        leaf = k.rsplit(".", 2)
        is_norm = ".norm" in k or "conv_norm_out" in k or "group_norm" in k
        if k == "pos_embed":
            out[k] = dit_pos_embed(shp[2], int(round(shp[1] ** 0.5)))
            continue
        if k.endswith("adaLN_modulation.1.weight") or k.endswith("embedding_table.weight"):
            t = 0.3 * torch.randn(shp, generator=g) / (1.0 if k.endswith("table.weight") else math.sqrt(shp[1] / 8.0))
            out[k] = t.to(torch.float32).contiguous()
            continue
        if is_norm and k.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif is_norm and k.endswith(".bias"):
            t = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            gain = 1.4 if (k.endswith("to_q.weight") or k.endswith("to_k.weight")) else 1.0
            a = gain * math.sqrt(3.0 / fan_in)
            t = (torch.rand(shp, generator=g) * 2 - 1) * a
        out[k] = t.to(torch.float32).contiguous()
    return out


_SHARED_FILES = []          # files this process (rank 0) wrote for the other ranks
_CLEANUP_HOOKED = False
_T_IMPORT = __import__("time").time()


def cleanup_shared():
    """Rank 0, once every rank has loaded (after a barrier): remove the shared files (they sit in /dev/shm, i.e. in memory).
    Also runs at interpreter exit and on SIGTERM / SIGINT of rank 0, so a job that dies between the write and the barrier does
    not leave 2.8 GB of SD1.5 weights behind."""
    import os
    while _SHARED_FILES:
        try:
            os.remove(_SHARED_FILES.pop())
        except OSError:
            pass


def _hook_cleanup():
    global _CLEANUP_HOOKED
    if _CLEANUP_HOOKED:
        return
    _CLEANUP_HOOKED = True
    import atexit
    import signal
    atexit.register(cleanup_shared)
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            prev = signal.getsignal(sig)

            def handler(signum, frame, _prev=prev):
                cleanup_shared()
                if callable(_prev):
                    _prev(signum, frame)
                else:
                    signal.signal(signum, signal.SIG_DFL)
                    import os
                    os.kill(os.getpid(), signum)
            signal.signal(sig, handler)
        except (ValueError, OSError):           # not the main thread: atexit still runs
            pass


def launch_nonce() -> str:
    """What the ranks of ONE launch share and no other launch has: DSIM_LAUNCH_NONCE when the launcher sets it, else the parent
    process id (the torch.distributed.run agent, or parallel.spawn_ranks' supervisor, is the parent of every rank)."""
    import os
    return os.environ.get("DSIM_LAUNCH_NONCE") or f"ppid{os.getppid()}"


def shared_fingerprint(sd: Dict[str, torch.Tensor]) -> int:
    """A 63-bit content fingerprint of a state dict (names, shapes and the bytes of up to 64 Ki elements per tensor): ranks compare
    it once the process group is up (bench.py), so a rank that mapped anything but rank 0's weights stops the job."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        t = sd[k]
        h.update(k.encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.detach().reshape(-1)[:65536].contiguous().cpu().numpy().tobytes())
    return int.from_bytes(h.digest()[:8], "little") >> 1


def make_state_dict_shared(cfg, seed: int = 0, keys: Optional[Sequence[str]] = None, rank: int = 0, world: int = 1,
                           cache_dir: Optional[str] = None, timeout_s: float = 900.0) -> Dict[str, torch.Tensor]:
    """``make_state_dict`` for the N ranks of ONE node: rank 0 synthesises the tensors (698 M parameters for SD1.5 to the tap:
    tens of seconds of host time) and writes them once as a safetensors file under `cache_dir` (default: /dev/shm or the temp
    directory); the other ranks wait for the file and map it instead of each repeating the synthesis on the same cores.  File
    based, so it works before any process group exists.  The name carries the config, seed, key set, the user id and the LAUNCH
    (`launch_nonce`); rank 0 ALWAYS writes (temporary name, mode 0600, then an atomic replace over whatever was there), and the
    other ranks accept only a file they own -- same uid, not group / world writable -- that was written after this launch began
    (a leftover of an earlier run, or a file another local user planted under the predictable name, is never loaded; they keep
    waiting for rank 0's).  Same tensors as make_state_dict, bit for bit; `shared_fingerprint` lets the ranks prove it."""
    if world <= 1:
        return make_state_dict(cfg, seed, keys)
    import hashlib
    import os
    import stat
    import tempfile
    import time
    from safetensors.torch import load_file, save_file
    d = cache_dir or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir())
    tag = hashlib.sha256((repr(cfg) + "|" + str(seed) + "|" + os.environ.get("MASTER_PORT", "") + "|" + launch_nonce() + "|" +
                          ",".join(sorted(keys) if keys is not None else ["*"])).encode()).hexdigest()[:16]
    path = os.path.join(d, f"dsim_synth_{os.getuid()}_{tag}.safetensors")
    if rank == 0:
        sd = make_state_dict(cfg, seed, keys)
        _hook_cleanup()
        tmp = f"{path}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        os.close(fd)
        _SHARED_FILES.append(tmp)
        save_file(sd, tmp)
        os.replace(tmp, path)
        _SHARED_FILES.remove(tmp)
        _SHARED_FILES.append(path)
        return sd
    t0 = time.monotonic()
    while True:
        try:
            st = os.stat(path)
            mine = st.st_uid == os.getuid() and not (st.st_mode & (stat.S_IWGRP | stat.S_IWOTH)) and stat.S_ISREG(st.st_mode)
            fresh = st.st_mtime >= _T_IMPORT - 120.0        # written by THIS launch's rank 0 (ranks start within seconds of each other)
            if mine and fresh:
                return load_file(path)
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: {path} did not appear within {timeout_s:.0f} s (did rank 0 fail?)")
        time.sleep(0.2)


def add_checkpoint_like_outliers(sd: Dict[str, torch.Tensor], seed: int = 5, conv_gain: float = 40.0, qk_gain: float = 5.0
                                 ) -> Dict[str, torch.Tensor]:
    """What trained checkpoints have and `make_state_dict` does not: (1) a few OUTLIER CHANNELS -- three output channels of
    every ResnetBlock2D conv scaled by `conv_gain` (weights and bias), so GroupNorm groups are dominated by one channel and
    the bf16 / fp16 range is exercised; (2) PEAKED attention -- the rows of one head of every self-attention to_q / to_k
    scaled by `qk_gain` (logits of that head x gain^2: near one-hot softmax rows, and at the 4096-key level scores far
    above key tile 0's maximum, i.e. the fixed-reference softmax's exact fallback, inside the U-Net).  Returns a modified copy."""
    out = dict(sd)
    g = torch.Generator("cpu").manual_seed(seed)
    for k in sorted(sd):
        t = sd[k]
        if (".resnets." in k and (k.endswith("conv1.weight") or k.endswith("conv2.weight"))) and t.ndim == 4:
            ch = torch.randperm(t.shape[0], generator=g)[:3]
            t = t.clone()
            t[ch] *= conv_gain
            out[k] = t
            b = k[:-len("weight")] + "bias"
            if b in sd:
                tb = sd[b].clone()
                tb[ch] *= conv_gain
                out[b] = tb
        elif ".attn1." in k and (k.endswith("to_q.weight") or k.endswith("to_k.weight")):
            t = t.clone()
            hd = 40 if t.shape[0] % 40 == 0 else 8            # one head's rows (SD1.5 heads are 40 / 80 / 160 wide)
            t[:hd] *= qk_gain
            out[k] = t
    return out


def make_heavy_tailed_latents(cfg: UNetConfig, pair_index: int, base_seed: int = 4321) -> Tuple[torch.Tensor, torch.Tensor]:
    """A latent pair with Student-t-like tails (a normal scaled by 1/sqrt(chi^2_3 / 3) per element, clipped at +-12):
    real VAE latents of saturated images are not Gaussian."""
    g = torch.Generator("cpu").manual_seed(base_seed + pair_index)
    s = cfg.sample_size
    zs = []
    for _ in range(2):
        z = torch.randn((1, cfg.in_channels, s, s), generator=g)
        chi = (torch.randn((3, 1, cfg.in_channels, s, s), generator=g) ** 2).sum(0) / 3.0
        zs.append((z / chi.sqrt().clamp_min(0.05)).clamp(-12.0, 12.0))
    return zs[0], zs[1]


def make_context(cfg: UNetConfig, seed: int = 77) -> torch.Tensor:
    """Stand-in for [CLIP(""), CLIP(prompt)] (diffsim_pipeline.py:125-141): (2, L, Dc) fp32."""
    g = torch.Generator("cpu").manual_seed(seed)
    return 0.5 * torch.randn((2, cfg.ctx_len, cfg.cross_attention_dim), generator=g)


def make_pooled(cfg: UNetConfig, seed: int = 78) -> torch.Tensor:
    """Stand-in for SDXL's pooled text embeddings [neg, pos]: (2, pooled_dim) fp32."""
    g = torch.Generator("cpu").manual_seed(seed)
    return 0.5 * torch.randn((2, cfg.pooled_dim), generator=g)


def _lowfreq(g: torch.Generator, ch: int, side: int, normal: bool) -> torch.Tensor:
    base = torch.randn((1, ch, 8, 8), generator=g) if normal else torch.rand((1, ch, 8, 8), generator=g)
    return torch.nn.functional.interpolate(base, size=(side, side), mode="bilinear",
                                           align_corners=False)


def make_pair_latents(cfg: UNetConfig, pair_index: int, base_seed: int = 1234
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Latents-in synthetic pair i: image A drawn first, then B; each (1,4,s,s) fp32,
    ``0.75*lowfreq + 0.65*randn`` -- unit variance like real SD latents after the 0.18215
    scaling (SURVEY.md section 8d proposed 0.18215*(...), whose std of 0.13 is drowned by the
    t=401 noise and gives no score spread between pairs)."""
    g = torch.Generator("cpu").manual_seed(base_seed + pair_index)
    s = cfg.sample_size
    zs = []
    for _ in range(2):
        lf = _lowfreq(g, cfg.in_channels, s, normal=True)
        hf = torch.randn((1, cfg.in_channels, s, s), generator=g)
        zs.append(0.75 * lf + 0.65 * hf)
    return zs[0], zs[1]


def draw_pair_noise(seed: int, shape: Sequence[int]) -> List[torch.Tensor]:
    """The four draws one reference call makes on its single generator, in order:
    vae-sample(A), vae-sample(B), noise(A), noise(B)
    (reference: diffsim/diffsim.py:109-113 and diffsim/diffsim_pipeline.py:174-176)."""
    g = torch.Generator("cpu").manual_seed(int(seed))
    return [torch.randn(tuple(shape), generator=g, dtype=torch.float32) for _ in range(4)]


def make_image_pair(pair_index: int, size: int = 512, base_seed: int = 1234) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pixels-in synthetic pair i: uint8-quantised ``0.5*noise + 0.5*lowfreq``
    images, returned as process_image-style (1,3,S,S) f32 tensors in [-1,1]; image A drawn first."""
    g = torch.Generator("cpu").manual_seed(base_seed + pair_index)
    outs = []
    for _ in range(2):
        hf = torch.randint(0, 256, (1, 3, size, size), generator=g).float()
        lf = _lowfreq(g, 3, size, normal=False) * 255.0
        img = torch.round(0.5 * hf + 0.5 * lf).clamp(0, 255) / 255.0
        outs.append((img - 0.5) / 0.5)
    return outs[0], outs[1]


def dit_pos_embed(dim: int, grid: int) -> torch.Tensor:
    """2-D sin/cos positional table of DiT (DiT/modelsdit.py:278-325; a fixed buffer, part of the state dict)."""
    import numpy as np

    def one_d(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        o = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(o), np.cos(o)], axis=1)
    gh = np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gh, gh), axis=0).reshape(2, 1, grid, grid)
    emb = np.concatenate([one_d(dim // 2, g[0]), one_d(dim // 2, g[1])], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0).contiguous()
