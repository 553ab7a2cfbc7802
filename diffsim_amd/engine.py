"""Python handle over the C ABI: U-Net-to-tap engine, fused score tail and single-op entry points.

PyTorch is used only as the owner of device memory and streams; all arithmetic happens in
libdiffsim_amd.so.  Every call passes ``tensor.data_ptr()`` and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .config import UNetConfig

_TORCH2DSIM = {torch.float32: _lib.DSIM_F32, torch.bfloat16: _lib.DSIM_BF16, torch.float16: _lib.DSIM_F16}


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DsimError("tensor is not on the GPU: the engine has no CPU path")
        if t is not None and not t.is_contiguous():
            raise _lib.DsimError("tensor must be contiguous")


def resolve_tap(cfg: UNetConfig, target_block: str, target_layer):
    """(absolute block index, attention index, transformer-block index) of the hooked attn1.
    SD1.5 (diffsim/diffsim.py:122-145): down_blocks[:-1][l] / mid / up_blocks[1:][l], always attentions[-1]
    .transformer_blocks[-1].  SDXL (diffsim/diffsim_xl.py:88-107): target_layer = [block, attention, tfm_block]
    into down_blocks[1:] / up_blocks[:-1]; mid = [attention, tfm_block]."""
    if not cfg.sdxl_tap:
        l = int(target_layer)
        if target_block == "down_blocks":
            return l, -1, -1
        if target_block == "mid_blocks":
            return 0, -1, -1
        return l + 1, -1, -1
    tl = [int(v) for v in target_layer]
    if target_block == "down_blocks":
        return tl[0] + 1, tl[1], tl[2]
    if target_block == "mid_blocks":
        return 0, tl[0], tl[1]
    return tl[0], tl[1], tl[2]


def _cfg_struct(cfg: UNetConfig, dtype: torch.dtype, tap_block: str, tap_layer) -> _lib.UNetCfgC:
    c = _lib.UNetCfgC()
    n = len(cfg.block_out_channels)
    c.in_channels, c.n_levels = cfg.in_channels, n
    for i in range(n):
        c.block_out_channels[i] = cfg.block_out_channels[i]
        c.down_has_attn[i] = int(cfg.down_block_types[i] == "CrossAttnDownBlock2D")
        c.up_has_attn[i] = int(cfg.up_block_types[i] == "CrossAttnUpBlock2D")
        c.heads_per_level[i] = cfg.heads(i) if cfg.heads_per_level else 0
        c.depth_per_level[i] = cfg.depth(i)
    c.layers_per_block = cfg.layers_per_block
    c.num_heads = cfg.num_attention_heads
    c.cross_attention_dim = cfg.cross_attention_dim
    c.norm_num_groups = cfg.norm_num_groups
    c.norm_eps = cfg.norm_eps
    c.sample_size = cfg.sample_size
    c.ctx_len = cfg.ctx_len
    c.compute_dtype = _TORCH2DSIM[dtype]
    c.tap_block = _lib.TAP[tap_block]
    c.tap_layer, c.tap_attn, c.tap_tfm = resolve_tap(cfg, tap_block, tap_layer)
    c.addition_embed = int(cfg.addition_embed)
    c.addition_time_embed_dim = cfg.addition_time_embed_dim
    c.pooled_dim = cfg.pooled_dim
    return c


class UNetEngine:
    """One handle = one (config, compute dtype, tap) triple with its own packed weights.

    Replaces ``self.unet(...)`` + the attention pre-hook of the reference
    (diffsim/diffsim_pipeline.py:213-221, diffsim/diffsim.py:43-56, 122-145).
    """

    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], dtype: torch.dtype = torch.bfloat16,
                 target_block: str = "up_blocks", target_layer: int = 0, device: str = "cuda:0"):
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("compute dtype must be float32 (parity mode), bfloat16 or float16")
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.DsimError("no GPU visible: the DiffSim engine runs only on the HIP device")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.target_block, self.target_layer = target_block, target_layer
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            cs = _cfg_struct(cfg, dtype, target_block, target_layer)
            _lib.check(self.L.dsim_unet_create(C.byref(cs), C.byref(self._h)), "dsim_unet_create")
            keep = []
            for k, v in state_dict.items():
                t = v.detach()
                if t.dtype not in _TORCH2DSIM:
                    t = t.float()
                t = t.to(self.device).contiguous()
                keep.append(t)
                shp = (C.c_int64 * t.ndim)(*t.shape)
                _lib.check(self.L.dsim_unet_load_weight(self._h, k.encode(), t.data_ptr(), _TORCH2DSIM[t.dtype],
                                                        shp, t.ndim), f"load_weight({k})")
            torch.cuda.synchronize(self.device)
            _lib.check(self.L.dsim_unet_finalize(self._h, _stream_ptr()), "dsim_unet_finalize")
            del keep
        self.sample_size = cfg.sample_size
        self._refresh_tap_shape()
        self._ws_by_stream: Dict[int, torch.Tensor] = {}      # one arena per HIP stream the engine is driven from
        self._graphs: Dict[tuple, tuple] = {}
        self.use_graphs = False
        self._profiling = False
        self._t = None

    def _refresh_tap_shape(self):
        n, h, d = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.L.dsim_unet_tap_shape(self._h, C.byref(n), C.byref(h), C.byref(d)), "tap_shape")
        self.tokens, self.heads, self.head_dim = n.value, h.value, d.value
        self._max_images = None

    def set_tap(self, target_block: str, target_layer):
        """Move the tap; the packed weights are shared (one copy per dtype serves every --target_block /
        --target_layer).  Raises if a parameter needed before the new tap was never loaded."""
        if (target_block, target_layer) == (self.target_block, self.target_layer):
            return
        tl, ta, tt = resolve_tap(self.cfg, target_block, target_layer)
        _lib.check(self.L.dsim_unet_set_tap(self._h, _lib.TAP[target_block], tl, ta, tt), "dsim_unet_set_tap")
        self.target_block, self.target_layer = target_block, target_layer
        self._refresh_tap_shape()           # (captured hipGraphs are keyed by tap and latent side: a sweep that alternates
                                            #  between taps keeps both graphs instead of re-capturing at every switch)

    def set_sample_size(self, side: int):
        """Latent side of the next qkv() calls (cfg.sample_size is the default, not a limit)."""
        if side != self.sample_size:
            # any side the reference accepts (--image_size a multiple of 8, argprocess.py:8): sides that are not a multiple of
            # 2**(levels-1) take the ceil-div stride-2 convs and the explicit-size upsample (diffusers' forward_upsample_size)
            if side < 2:
                raise ValueError(f"latent side {side}: --image_size must be at least 16")
            _lib.check(self.L.dsim_unet_set_sample_size(self._h, int(side)), "dsim_unet_set_sample_size")
            self.sample_size = int(side)
            self._refresh_tap_shape()

    def set_cfg_dedup(self, enable: bool):
        """Opt-in: compute the part of the graph both CFG halves share once (SD1.5 graphs; bit-identical scores)."""
        _lib.check(self.L.dsim_unet_set_cfg_dedup(self._h, int(bool(enable))), "dsim_unet_set_cfg_dedup")
        self._max_images = None
        self._graphs.clear()

    def set_fusion(self, mask: int):
        """Which multi-operator kernels replace their unfused chains (_lib.FUSE_* bits; default all).  0 = every layer
        its own launch: the A/B switch of bench.py --fusion and of the parity tests."""
        _lib.check(self.L.dsim_unet_set_fusion(self._h, int(mask)), "dsim_unet_set_fusion")
        self._max_images = None
        self._graphs.clear()

    def view(self, target_block: str, target_layer) -> "TapView":
        return TapView(self, target_block, target_layer)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.L.dsim_unet_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_timestep(self, t: int):
        if self._t != t:
            with torch.cuda.device(self.device):
                _lib.check(self.L.dsim_unet_set_timestep(self._h, int(t), _stream_ptr()), "set_timestep")
            self._t = t
            self._graphs.clear()

    def set_conditioning(self, t: int, text_embeds: torch.Tensor, time_ids: torch.Tensor):
        """SDXL: timestep + added conditioning (pooled text embeds (2,P) [neg,pos], time ids (2,6))."""
        te = text_embeds.to(self.device, torch.float32).contiguous()
        ti = time_ids.to(self.device, torch.float32).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.L.dsim_unet_set_conditioning(self._h, int(t), te.data_ptr(), ti.data_ptr(), _stream_ptr()),
                       "set_conditioning")
            torch.cuda.synchronize(self.device)     # te/ti may be freed by the caller
        self._t = ("cond", t)
        self._graphs.clear()

    def profile(self, enable: bool):
        self._profiling = bool(enable)
        _lib.check(self.L.dsim_unet_profile(self._h, int(enable)), "profile")

    def profile_records(self, detail: bool = False):
        """[(kernel family, algorithmic flops, algorithmic bytes, ms)] of the forwards run since
        profile(True); synchronises the device first.  detail=True appends the launch's shape string."""
        torch.cuda.synchronize(self.device)
        out = []
        buf = C.create_string_buffer(160)
        fl, by, ms = C.c_double(), C.c_double(), C.c_double()
        for i in range(self.L.dsim_unet_profile_count(self._h)):
            _lib.check(self.L.dsim_unet_profile_get(self._h, i, buf, 160, C.byref(fl), C.byref(by), C.byref(ms)),
                       "profile_get")
            fam, _, shape = buf.value.decode().partition("|")
            out.append((fam, fl.value, by.value, ms.value, shape) if detail else (fam, fl.value, by.value, ms.value))
        return out

    def workspace_bytes(self, n_images: int) -> int:
        return int(self.L.dsim_unet_workspace_bytes(self._h, n_images))

    def max_images(self, upper: int = 4096) -> int:
        """Largest n_images one qkv() call accepts (every activation < 2 GiB); bisection over the dry-run planner,
        cached (it depends on the graph only)."""
        if getattr(self, "_max_images", None) is not None:
            return self._max_images
        self._max_images = self._max_images_search(upper)
        return self._max_images

    def _max_images_search(self, upper: int) -> int:
        if self.workspace_bytes(1) == 0:
            return 0
        lo, hi = 1, upper
        if self.workspace_bytes(hi):
            return hi
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if self.workspace_bytes(mid):
                lo = mid
            else:
                hi = mid
        return lo

    def qkv(self, latents: torch.Tensor, noise: torch.Tensor, sqrt_abar: float, sqrt_1m_abar: float,
            ctx: torch.Tensor, out: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
        """latents/noise (n,Cin,s,s) f32 cuda; ctx (2,L,Dc) f32 cuda -> q,k,v each
        [n][2][tokens][heads*head_dim] in the compute dtype."""
        _require_cuda(latents, noise, ctx)
        if latents.dtype != torch.float32 or noise.dtype != torch.float32 or ctx.dtype != torch.float32:
            raise _lib.DsimError("latents, noise and ctx must be float32")
        n = latents.shape[0]
        if latents.ndim != 4 or latents.shape[1] != self.cfg.in_channels or latents.shape[2] != latents.shape[3] or \
                noise.shape != latents.shape:
            raise _lib.DsimError(f"latents and noise must be (n,{self.cfg.in_channels},s,s)")
        self.set_sample_size(int(latents.shape[2]))
        if tuple(ctx.shape) != (2, self.cfg.ctx_len, self.cfg.cross_attention_dim):
            raise _lib.DsimError("ctx must be (2, ctx_len, cross_attention_dim)")
        with torch.cuda.device(self.device):
            need = self.workspace_bytes(n)
            if need == 0:
                raise _lib.DsimError(f"{n} images do not fit one call (an activation would reach 2 GiB): at most "
                                     f"{self.max_images()} images per call for this graph")
            # dsim_unet_qkv keeps no per-call state in the handle, so independent batches may be in flight on several
            # streams at once (one host thread): each stream gets its own workspace arena
            sid = _stream_ptr()
            ws = self._ws_by_stream.get(sid)
            if ws is None or ws.numel() < need:
                self._ws_by_stream.pop(sid, None)
                self._graphs.clear()                 # captured graphs hold the old arena's addresses
                ws = self._ws_by_stream[sid] = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws = ws
            shape = (n, 2, self.tokens, self.heads * self.head_dim)
            if self.use_graphs and out is None and not self._profiling:
                return self._replay(latents, noise, float(sqrt_abar), float(sqrt_1m_abar), ctx, shape)
            if out is None:
                out = tuple(torch.empty((3,) + tuple(shape), dtype=self.dtype, device=self.device).unbind(0))    # one allocation: the tapped q | k | v projection is then one launch
            self._launch(latents, noise, float(sqrt_abar), float(sqrt_1m_abar), ctx, out)
        return out

    def _launch(self, latents, noise, sa, sb, ctx, out):
        q, k, v = out
        _lib.check(self.L.dsim_unet_qkv(self._h, latents.data_ptr(), noise.data_ptr(), sa, sb, ctx.data_ptr(),
                                        latents.shape[0], q.data_ptr(), k.data_ptr(), v.data_ptr(), self._ws.data_ptr(),
                                        self._ws.numel(), _stream_ptr()), "dsim_unet_qkv")

    def _replay(self, latents, noise, sa, sb, ctx, shape):
        """hipGraph path for launch-bound small batches: the ~330 kernel launches of one forward are captured
        once per (n_images, sqrt_abar, sqrt_1m_abar) over static input/output buffers and replayed as one graph
        launch.  dsim_unet_qkv never allocates or synchronises, so plain stream capture works."""
        key = (self.target_block, str(self.target_layer), self.sample_size, _stream_ptr(), shape[0], sa, sb)
        ent = self._graphs.get(key)
        if ent is None:
            st = {"lat": torch.empty_like(latents), "nz": torch.empty_like(noise), "ctx": torch.empty_like(ctx),
                  "out": tuple(torch.empty((3,) + tuple(shape), dtype=self.dtype, device=self.device).unbind(0))}
            st["lat"].copy_(latents), st["nz"].copy_(noise), st["ctx"].copy_(ctx)
            self._launch(st["lat"], st["nz"], sa, sb, st["ctx"], st["out"])      # eager warm-up (code objects loaded)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch(st["lat"], st["nz"], sa, sb, st["ctx"], st["out"])
            ent = self._graphs[key] = (g, st)
        g, st = ent
        st["lat"].copy_(latents), st["nz"].copy_(noise), st["ctx"].copy_(ctx)
        g.replay()
        return tuple(t.clone() for t in st["out"])


class TapView:
    """One (target_block, target_layer) of a shared UNetEngine: every attribute access first moves the engine's tap
    there, so several taps can be used alternately over ONE packed weight copy."""

    def __init__(self, base: UNetEngine, target_block: str, target_layer):
        object.__setattr__(self, "_base", base)
        object.__setattr__(self, "_tap", (target_block, target_layer))

    def __getattr__(self, name):
        base = object.__getattribute__(self, "_base")
        base.set_tap(*object.__getattribute__(self, "_tap"))
        return getattr(base, name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_base"), name, value)


def pair_score(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, idx_a: torch.Tensor, idx_b: torch.Tensor,
               heads: int, similarity: str = "cosine", return_status: bool = False):
    """Fused score tail (diffsim/diffsim.py:177-197).  q,k,v: [n_feat][B][N][H*D]; idx: int32 cuda [n_pairs].
    return_status: also return an int32 [n_pairs] tensor, 1 where the score is NaN / infinite (NaN guard)."""
    L = _lib.lib()
    _require_cuda(q, k, v, idx_a, idx_b)
    if similarity not in ("cosine", "mse"):
        raise ValueError(similarity)
    if q.dtype not in (torch.float32, torch.bfloat16, torch.float16) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise _lib.DsimError("q,k,v must share dtype float32, bfloat16 or float16")
    if idx_a.dtype != torch.int32 or idx_b.dtype != torch.int32:
        raise _lib.DsimError("pair indices must be int32")
    nf, B, N, HD = q.shape
    D = HD // heads
    n_pairs = idx_a.numel()
    out = torch.empty(n_pairs, dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        wsb = int(L.dsim_pair_score_workspace_bytes(n_pairs, B, heads, N, D))
        ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
        if return_status:
            status = torch.empty(n_pairs, dtype=torch.int32, device=q.device)
            _lib.check(L.dsim_pair_score_status(q.data_ptr(), k.data_ptr(), v.data_ptr(), idx_a.data_ptr(), idx_b.data_ptr(),
                                                n_pairs, B, heads, N, D, _TORCH2DSIM[q.dtype],
                                                0 if similarity == "cosine" else 1, out.data_ptr(), status.data_ptr(),
                                                ws.data_ptr(), wsb, _stream_ptr()), "dsim_pair_score_status")
            return out, status
        _lib.check(L.dsim_pair_score(q.data_ptr(), k.data_ptr(), v.data_ptr(), idx_a.data_ptr(), idx_b.data_ptr(),
                                     n_pairs, B, heads, N, D, _TORCH2DSIM[q.dtype], 0 if similarity == "cosine" else 1,
                                     out.data_ptr(), ws.data_ptr(), wsb, _stream_ptr()), "dsim_pair_score")
    return out


# ---- single-operator entry points (kernel-level parity tests) -----------------------------------
def op_linear(x, w, bias=None, residual=None, geglu=False):
    L = _lib.lib()
    _require_cuda(x, w, bias, residual)
    M, K = x.shape
    N = w.shape[0] // 2 if geglu else w.shape[0]
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _lib.check(L.dsim_op_linear(x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), M, N, K,
                                _TORCH2DSIM[x.dtype], int(geglu), _stream_ptr()), "op_linear")
    return out


def op_ff_fused(x, ln_g, ln_b, w1, b1, w2, b2, eps=1e-5):
    """x + ff.net.2(GEGLU(ff.net.0.proj(LayerNorm(x)))) in one launch (bf16, C = 320); w1 [8C][C], w2 [C][4C] f32."""
    L = _lib.lib()
    _require_cuda(x, ln_g, ln_b, w1, b1, w2, b2)
    if x.dtype != torch.bfloat16:          # the op-level entry point packs and launches in bf16 only (the engines run the fp16 twin)
        raise TypeError("op_ff_fused takes a torch.bfloat16 input")
    M, Cc = x.shape
    out = torch.empty_like(x)
    _lib.check(L.dsim_op_ff_fused(x.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                  b2.data_ptr(), out.data_ptr(), M, Cc, float(eps), _stream_ptr()), "op_ff_fused")
    return out


def op_ln_linear(x, ln_g, ln_b, w, eps=1e-5):
    """LayerNorm(x) W^T in one launch (bf16, C = 320, bias-free, N a multiple of 64 up to 960); w [N][C] f32; ln_g = ln_b = None
    skips the LayerNorm."""
    L = _lib.lib()
    _require_cuda(x, ln_g, ln_b, w)
    if x.dtype != torch.bfloat16:          # as op_ff_fused: bf16 only at the op level
        raise TypeError("op_ln_linear takes a torch.bfloat16 input")
    M, Cc = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _lib.check(L.dsim_op_ln_linear(x.data_ptr(), _ptr(ln_g), _ptr(ln_b), w.data_ptr(), out.data_ptr(), M, Cc, N, float(eps),
                                   _stream_ptr()), "op_ln_linear")
    return out


def op_conv3x3(x, w, bias=None, residual=None, stride=1, upsample=False):
    """x: [B][H][W][Cin] token-major; w: [Cout][Cin][3][3] f32."""
    L = _lib.lib()
    _require_cuda(x, w, bias, residual)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (2 * H, 2 * W) if upsample else (((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W))      # ceil: odd sides (as the executor)
    out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    _lib.check(L.dsim_op_conv3x3(x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), B, H, W, Cin,
                                 Cout, stride, int(upsample), _TORCH2DSIM[x.dtype], _stream_ptr()), "op_conv3x3")
    return out


def op_groupnorm(x0, x1, gamma, beta, groups, eps, silu):
    """x0: [B][HW][C0], x1: optional [B][HW][C1] (channel concat)."""
    L = _lib.lib()
    _require_cuda(x0, x1, gamma, beta)
    B, HW, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[2]
    out = torch.empty((B, HW, C0 + C1), dtype=x0.dtype, device=x0.device)
    _lib.check(L.dsim_op_groupnorm(x0.data_ptr(), C0, _ptr(x1), C1, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                   B, HW, groups, float(eps), int(silu), _TORCH2DSIM[x0.dtype], _stream_ptr()),
               "op_groupnorm")
    return out


def op_layernorm(x, gamma, beta, eps=1e-5):
    L = _lib.lib()
    _require_cuda(x, gamma, beta)
    M, Cc = x.shape
    out = torch.empty_like(x)
    _lib.check(L.dsim_op_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), M, Cc, float(eps),
                                   _TORCH2DSIM[x.dtype], _stream_ptr()), "op_layernorm")
    return out


def op_attention(q, k, v, heads, fp8: bool = False):
    """q: [B][Nq][H*D]; k,v: [Bkv][Nk][H*D] -> [B][Nq][H*D].  fp8: bf16 tensors, e4m3 MFMAs (DiT config 5)."""
    L = _lib.lib()
    _require_cuda(q, k, v)
    B, Nq, HD = q.shape
    Bkv, Nk, _ = k.shape
    out = torch.empty_like(q)
    if fp8:
        if q.dtype != torch.bfloat16:
            raise _lib.DsimError("fp8 attention takes bf16 tensors")
        _lib.check(L.dsim_op_attention_fp8(q.data_ptr(), HD, k.data_ptr(), v.data_ptr(), HD, out.data_ptr(), HD, B, Bkv, heads,
                                           Nq, Nk, HD // heads, _stream_ptr()), "op_attention_fp8")
        return out
    _lib.check(L.dsim_op_attention(q.data_ptr(), HD, k.data_ptr(), v.data_ptr(), HD, out.data_ptr(), HD, B, Bkv, heads,
                                   Nq, Nk, HD // heads, _TORCH2DSIM[q.dtype], _stream_ptr()), "op_attention")
    return out


# ---- the arithmetic either side of the VAE encoder, on the device ------------------------------------------------
def image_preprocess(pixels_u8: torch.Tensor, to_half: bool = False) -> torch.Tensor:
    """pixels u8 cuda [n][H][W][3] (decoded + Lanczos-resized on the host) -> process_image's f32 [n][3][H][W], bit-identical
    to the numpy arithmetic of /root/reference/diffsim/diffsim.py:31-41; to_half rounds through fp16 (diffsim.py:93)."""
    L = _lib.lib()
    _require_cuda(pixels_u8)
    if pixels_u8.dtype != torch.uint8 or pixels_u8.ndim != 4 or pixels_u8.shape[3] != 3:
        raise _lib.DsimError("pixels must be uint8 [n][H][W][3]")
    n, H, W, _ = pixels_u8.shape
    out = torch.empty((n, 3, H, W), dtype=torch.float32, device=pixels_u8.device)
    _lib.check(L.dsim_image_preprocess(pixels_u8.contiguous().data_ptr(), out.data_ptr(), n, H, W, int(to_half), _stream_ptr()),
               "dsim_image_preprocess")
    return out


def latent_sample(moments: torch.Tensor, eps: torch.Tensor, scaling_factor: float, first: int = 0, stride: int = 1,
                  round_fp16: bool = False) -> torch.Tensor:
    """scaling_factor * DiagonalGaussianDistribution(moments).sample() with the caller's draw `eps` ((1,C,h,w) shared or
    (n_out,C,h,w)), for images first, first+stride, ... of `moments` ((n,2C,h,w) f32 cuda) -> (n_out,C,h,w) f32."""
    L = _lib.lib()
    moments, eps = moments.contiguous(), eps.contiguous()
    _require_cuda(moments, eps)
    n, C2, h, w = moments.shape
    Cc = C2 // 2
    n_out = (n - first + stride - 1) // stride
    if moments.dtype != torch.float32 or eps.dtype != torch.float32 or eps.shape[1:] != (Cc, h, w) or eps.shape[0] not in (1, n_out):
        raise _lib.DsimError("latent_sample: moments (n,2C,h,w) f32, eps (1|n_out,C,h,w) f32")
    out = torch.empty((n_out, Cc, h, w), dtype=torch.float32, device=moments.device)
    _lib.check(L.dsim_latent_sample(moments.data_ptr(), eps.data_ptr(), out.data_ptr(), n_out, first,
                                    stride, Cc, h * w, eps.shape[0], float(scaling_factor), int(round_fp16), _stream_ptr()),
               "dsim_latent_sample")
    return out


# ---- VAE encoder ------------------------------------------------------------
class _LatentDist:
    """``DiagonalGaussianDistribution`` surface the reference uses: ``.sample(generator)``
    (diffsim/diffsim.py:94).  The moments live on the device; the noise is drawn with the caller's
    generator on ITS device (CPU in the reference-CPU-path setting) in the reference's order, and
    mean + exp(0.5 clamp(logvar)) * eps is one launch of dsim_latent_sample -- the same arithmetic the batched paths use,
    so a per-pair call and a chunked run give bit-identical latents."""

    def __init__(self, moments: torch.Tensor, sample_dtype: torch.dtype = torch.float32):
        self.moments = moments
        # dtype of the sample draw: diffusers draws randn_tensor(dtype=parameters.dtype), i.e. fp16 under the
        # reference's fp16 SD1.5 pipeline -- a different random stream from the fp32 draw of the same generator
        self.sample_dtype = sample_dtype

    @property
    def mean(self) -> torch.Tensor:
        return self.moments.chunk(2, dim=1)[0]

    @property
    def logvar(self) -> torch.Tensor:
        return self.moments.chunk(2, dim=1)[1].clamp(-30.0, 20.0)

    @property
    def std(self) -> torch.Tensor:
        return torch.exp(0.5 * self.logvar)

    def sample(self, generator=None) -> torch.Tensor:
        gdev = generator.device if generator is not None else self.moments.device
        eps = torch.randn(self.mean.shape, generator=generator, dtype=self.sample_dtype, device=gdev)
        return latent_sample(self.moments.float(), eps.to(self.moments.device, torch.float32), 1.0)

    def mode(self) -> torch.Tensor:
        return self.mean


class _EncodeOut:
    def __init__(self, moments, sample_dtype=torch.float32):
        self.latent_dist = _LatentDist(moments, sample_dtype)


class VAEEncoder:
    """``AutoencoderKL.encode`` on the HIP engine, with the surface DiffSim.prepare_image_latents needs:
    ``vae.encode(image).latent_dist.sample(generator)`` and ``vae.config.scaling_factor``."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], dtype: torch.dtype = torch.bfloat16,
                 device: str = "cuda:0"):
        import types
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.DsimError("no GPU visible: the VAE encoder runs only on the HIP device")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.config = types.SimpleNamespace(scaling_factor=cfg.scaling_factor)
        c = _lib.VAECfgC()
        c.in_channels, c.latent_channels, c.n_levels = cfg.in_channels, cfg.latent_channels, len(cfg.block_out_channels)
        for i, v in enumerate(cfg.block_out_channels):
            c.block_out_channels[i] = v
        c.layers_per_block, c.norm_num_groups = cfg.layers_per_block, cfg.norm_num_groups
        c.compute_dtype = _TORCH2DSIM[dtype]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.dsim_vae_create(C.byref(c), C.byref(self._h)), "dsim_vae_create")
            keep = []
            for k, v in state_dict.items():
                if not (k.startswith("encoder.") or k.startswith("quant_conv.")):
                    continue
                t = v.detach()
                if t.dtype not in _TORCH2DSIM:
                    t = t.float()
                t = t.to(self.device).contiguous()
                keep.append(t)
                shp = (C.c_int64 * t.ndim)(*t.shape)
                _lib.check(self.L.dsim_vae_load_weight(self._h, k.encode(), t.data_ptr(), _TORCH2DSIM[t.dtype], shp, t.ndim),
                           f"vae load_weight({k})")
            torch.cuda.synchronize(self.device)
            _lib.check(self.L.dsim_vae_finalize(self._h, _stream_ptr()), "dsim_vae_finalize")
            del keep
        self._ws = None
        self.sample_dtype = torch.float32      # dtype of latent_dist.sample's draw (DiffSim(noise_dtype=...) sets it)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.L.dsim_vae_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def moments(self, images: torch.Tensor) -> torch.Tensor:
        """images (n,3,S,S) in [-1,1], any float dtype/device -> moments (n, 2*latent, S/8, S/8) f32 on device."""
        x = images.to(self.device).float().contiguous()      # an fp16 image keeps its fp16-rounded pixel values
        n, cin, S, S2 = x.shape
        if cin != self.cfg.in_channels or S != S2:
            raise _lib.DsimError("images must be (n, in_channels, S, S)")
        f = 2 ** (len(self.cfg.block_out_channels) - 1)
        out = torch.empty((n, 2 * self.cfg.latent_channels, S // f, S // f), dtype=torch.float32, device=self.device)
        # the kernels address activations through 32-bit buffer offsets: keep every tensor < 2 GiB
        es = 4 if self.dtype == torch.float32 else 2
        chunk = max(1, (2 ** 30) // (S * S * max(self.cfg.block_out_channels[0], 1) * es))
        with torch.cuda.device(self.device):
            for i0 in range(0, n, chunk):
                m = min(chunk, n - i0)
                need = int(self.L.dsim_vae_workspace_bytes(self._h, m, S))
                if need == 0:
                    raise _lib.DsimError("unsupported image size for the VAE encoder")
                if self._ws is None or self._ws.numel() < need:
                    self._ws = None
                    self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                _lib.check(self.L.dsim_vae_encode(self._h, x[i0:i0 + m].data_ptr(), m, S, out[i0:i0 + m].data_ptr(),
                                                  self._ws.data_ptr(), self._ws.numel(), _stream_ptr()), "dsim_vae_encode")
        return out

    def encode(self, images: torch.Tensor) -> _EncodeOut:
        return _EncodeOut(self.moments(images), self.sample_dtype)

    def profile(self, enable: bool):
        """HIP-event brackets around every launch of the following encodes (dsim_vae_profile); not inside a timed region."""
        _lib.check(self.L.dsim_vae_profile(self._h, int(enable)), "vae profile")

    def profile_records(self, detail: bool = False):
        """Same records as UNetEngine.profile_records."""
        torch.cuda.synchronize(self.device)
        out = []
        buf = C.create_string_buffer(160)
        fl, by, ms = C.c_double(), C.c_double(), C.c_double()
        for i in range(self.L.dsim_vae_profile_count(self._h)):
            _lib.check(self.L.dsim_vae_profile_get(self._h, i, buf, 160, C.byref(fl), C.byref(by), C.byref(ms)), "vae profile_get")
            fam, _, shape = buf.value.decode().partition("|")
            out.append((fam, fl.value, by.value, ms.value, shape) if detail else (fam, fl.value, by.value, ms.value))
        return out


# ---- DiT backbone ---------------------------------------------------------
class DiTEngine:
    """One handle = (DiT config, compute dtype, tapped block)."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], dtype: torch.dtype = torch.bfloat16, target_layer: int = 0,
                 device: str = "cuda:0"):
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.DsimError("no GPU visible: the DiT engine runs only on the HIP device")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        c = _lib.DiTCfgC()
        for f in ("input_size", "patch_size", "in_channels", "hidden_size", "depth", "num_heads", "mlp_ratio", "num_classes",
                  "freq_dim"):
            setattr(c, f, getattr(cfg, f))
        c.compute_dtype, c.tap_layer = _TORCH2DSIM[dtype], int(target_layer)
        self.target_layer = int(target_layer)
        self.tokens = (cfg.input_size // cfg.patch_size) ** 2
        self.heads, self.head_dim = cfg.num_heads, cfg.hidden_size // cfg.num_heads
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.dsim_dit_create(C.byref(c), C.byref(self._h)), "dsim_dit_create")
            keep = []
            for k, v in state_dict.items():
                if k.startswith("final_layer"):
                    continue
                t = v.detach()
                if t.dtype not in _TORCH2DSIM:
                    t = t.float()
                t = t.to(self.device).contiguous()
                keep.append(t)
                shp = (C.c_int64 * t.ndim)(*t.shape)
                _lib.check(self.L.dsim_dit_load_weight(self._h, k.encode(), t.data_ptr(), _TORCH2DSIM[t.dtype], shp, t.ndim),
                           f"dit load_weight({k})")
            torch.cuda.synchronize(self.device)
            _lib.check(self.L.dsim_dit_finalize(self._h, _stream_ptr()), "dsim_dit_finalize")
            del keep
        self._ws = None
        self._cond = None

    def set_tap(self, layer: int):
        """Move the tapped block; the packed weights are shared by every --target_layer (dsim_dit_set_tap)."""
        if int(layer) != self.target_layer:
            _lib.check(self.L.dsim_dit_set_tap(self._h, int(layer)), "dsim_dit_set_tap")
            self.target_layer = int(layer)

    def set_attention(self, fp8: bool):
        """fp8 (OCP e4m3) MFMA attention in the DiT blocks (BASELINE config 5); bf16 handles only."""
        _lib.check(self.L.dsim_dit_set_attention(self._h, 1 if fp8 else 0), "dsim_dit_set_attention")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.L.dsim_dit_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def profile(self, enable: bool):
        _lib.check(self.L.dsim_dit_profile(self._h, int(enable)), "dit profile")

    def profile_records(self, detail: bool = False):
        """Same records as UNetEngine.profile_records."""
        torch.cuda.synchronize(self.device)
        out = []
        buf = C.create_string_buffer(160)
        fl, by, ms = C.c_double(), C.c_double(), C.c_double()
        for i in range(self.L.dsim_dit_profile_count(self._h)):
            _lib.check(self.L.dsim_dit_profile_get(self._h, i, buf, 160, C.byref(fl), C.byref(by), C.byref(ms)), "dit profile_get")
            fam, _, shape = buf.value.decode().partition("|")
            out.append((fam, fl.value, by.value, ms.value, shape) if detail else (fam, fl.value, by.value, ms.value))
        return out

    def set_conditioning(self, t_model: int, y0: int, y1: int):
        if self._cond != (t_model, y0, y1):
            with torch.cuda.device(self.device):
                _lib.check(self.L.dsim_dit_set_conditioning(self._h, int(t_model), int(y0), int(y1), _stream_ptr()),
                           "dit set_conditioning")
            self._cond = (t_model, y0, y1)

    def qkv(self, latents: torch.Tensor, noise: torch.Tensor, sa: float, sb: float):
        _require_cuda(latents, noise)
        n = latents.shape[0]
        s = self.cfg.input_size
        if tuple(latents.shape) != (n, self.cfg.in_channels, s, s) or latents.dtype != torch.float32 or noise.shape != latents.shape:
            raise _lib.DsimError(f"latents/noise must be float32 (n,{self.cfg.in_channels},{s},{s})")
        with torch.cuda.device(self.device):
            need = int(self.L.dsim_dit_workspace_bytes(self._h, n))
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            shape = (n, 2, self.tokens, self.cfg.hidden_size)
            q, k, v = (torch.empty(shape, dtype=self.dtype, device=self.device) for _ in range(3))
            _lib.check(self.L.dsim_dit_qkv(self._h, latents.data_ptr(), noise.data_ptr(), float(sa), float(sb), n, q.data_ptr(),
                                           k.data_ptr(), v.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _stream_ptr()),
                       "dsim_dit_qkv")
        return q, k, v
