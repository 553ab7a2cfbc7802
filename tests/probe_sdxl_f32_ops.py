#!/usr/bin/env python3
"""Diagnostic (CPU only, run on a many-core host): WHICH class of torch CPU fp32 kernels accounts for the 1.0e-4 (relative, on the
score) distance between the fp32 CPU oracle and the float64 evaluation of the same SDXL graph at 1024 px
(profiles/r02_sdxl_f64_probe.txt; the HIP fp32 mode is 2e-8 from float64)?  The oracle is run in fp32 with ONE operator class
at a time evaluated in float64 and rounded back to fp32 at its output: normalisations (GroupNorm + LayerNorm), matrix products
(Linear + Conv2d), attention (scaled_dot_product_attention).  The class whose upgrade closes the gap is the one that loses the
digits.  Test infrastructure (it runs the oracle), not collected by pytest.  Usage: python tests/probe_sdxl_f32_ops.py"""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsim_amd import config as C, synth as S          # noqa: E402
from oracle import cpu_ref as R                          # noqa: E402
from tests.test_gpu_fullsize import _oracle_unet, _qkv_at_taps   # noqa: E402


def upgrade(unet, which):
    """Evaluate one operator class in float64 (inputs and parameters cast up, output rounded to fp32)."""
    undo = []
    if which == "norms":
        for m in unet.modules():
            if isinstance(m, nn.GroupNorm):
                m.forward = (lambda x, m=m: F.group_norm(x.double(), m.num_groups, m.weight.double(), m.bias.double(), m.eps).float())
            elif isinstance(m, nn.LayerNorm):
                m.forward = (lambda x, m=m: F.layer_norm(x.double(), m.normalized_shape, None if m.weight is None else m.weight.double(),
                                                         None if m.bias is None else m.bias.double(), m.eps).float())
    elif which == "matmul":
        for m in unet.modules():
            if isinstance(m, nn.Linear):
                m.forward = (lambda x, m=m: F.linear(x.double(), m.weight.double(), None if m.bias is None else m.bias.double()).float())
            elif isinstance(m, nn.Conv2d):
                m.forward = (lambda x, m=m: F.conv2d(x.double(), m.weight.double(), None if m.bias is None else m.bias.double(), m.stride,
                                                     m.padding, m.dilation, m.groups).float())
    elif which == "attention":
        orig = F.scaled_dot_product_attention
        R.F.scaled_dot_product_attention = lambda q, k, v, *a, **kw: orig(q.double(), k.double(), v.double(), *a, **kw).float()
        undo.append(lambda: setattr(R.F, "scaled_dot_product_attention", orig))
    return undo


def main():
    cfg = C.SDXL
    drop = ("up_blocks.1", "up_blocks.2", "conv_norm_out", "conv_out", "up_blocks.0.attentions.1", "up_blocks.0.attentions.2",
            "up_blocks.0.resnets.1", "up_blocks.0.resnets.2", "up_blocks.0.upsamplers")
    shapes = C.unet_param_shapes(cfg)
    sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if not k.startswith(drop)])
    ctx, pooled = S.make_context(cfg), S.make_pooled(cfg)
    g = torch.Generator("cpu").manual_seed(1234)
    shp = (1, 4, 128, 128)
    zA, zB = torch.randn(shp, generator=g), torch.randn(shp, generator=g)
    n = S.draw_pair_noise(2334, shp)
    taps = {"a": ("up_blocks", [0, 0, 0])}
    res = {}
    for name, dt, which in (("f64", torch.float64, None), ("f32", torch.float32, None), ("f32 + norms in f64", torch.float32, "norms"),
                            ("f32 + Linear/Conv2d in f64", torch.float32, "matmul"), ("f32 + SDPA in f64", torch.float32, "attention")):
        unet = _oracle_unet(R, R.SDXL, sd, shapes, dt)
        undo = upgrade(unet, which) if which else []
        t0 = time.time()
        feats = []
        for z, nz in ((zA, n[2]), (zB, n[3])):
            x, t = R.sdxl_inputs(z, nz, 600)
            added = {"text_embeds": pooled.to(dt), "time_ids": R.sdxl_time_ids(unet.cfg).repeat(2, 1).to(dt)}
            feats.append(_qkv_at_taps(R, unet, torch.cat([x] * 2).to(dt), t, ctx.to(dt), added, taps))
        for u in undo:
            u()
        fa, fb = feats[0]["a"], feats[1]["a"]
        res[name] = float(R.pair_score(*[f.double() for f in fa], *[f.double() for f in fb], "cosine"))      # tail in float64
        if dt == torch.float32 and which is None:
            # the SAME fp32 features through the score tail (diffsim.py:177-197) in fp32, and with its pieces in float64
            qa, ka, va = fa
            qb, kb, vb = fb
            sd_ = F.scaled_dot_product_attention
            o32 = [sd_(qa, kb, vb), sd_(qa, ka, va), sd_(qb, ka, va), sd_(qb, kb, vb)]
            o64 = [sd_(qa.double(), kb.double(), vb.double()), sd_(qa.double(), ka.double(), va.double()),
                   sd_(qb.double(), ka.double(), va.double()), sd_(qb.double(), kb.double(), vb.double())]
            cs = lambda a, b: float(F.cosine_similarity(a.reshape(1, -1), b.reshape(1, -1)))
            for label, val in (("  fp32 features, whole tail in fp32 (= the fp32 CPU oracle)", 0.5 * (cs(o32[0], o32[1]) + cs(o32[2], o32[3]))),
                               ("  fp32 features, SDPA fp32, cosine in float64", 0.5 * (cs(o32[0].double(), o32[1].double()) + cs(o32[2].double(), o32[3].double()))),
                               ("  fp32 features, SDPA float64 -> fp32, cosine fp32", 0.5 * (cs(o64[0].float(), o64[1].float()) + cs(o64[2].float(), o64[3].float())))):
                print(f"{label:70s} score {val:.10f}   |x - f64| / f64 = {abs(val - res['f64']) / abs(res['f64']):.3e}  (elements per cosine: {o32[0].numel()})", flush=True)
        rel = abs(res[name] - res["f64"]) / abs(res["f64"])
        print(f"{name:32s} score {res[name]:.10f}   |x - f64| / f64 = {rel:.3e}   ({time.time() - t0:.0f} s, {torch.get_num_threads()} threads)", flush=True)
        del unet


if __name__ == "__main__":
    main()
