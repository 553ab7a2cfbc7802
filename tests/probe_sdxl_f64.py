#!/usr/bin/env python3
"""Diagnostic (GPU box): how far are (a) the fp32 HIP path and (b) the fp32 torch-CPU oracle from the SAME SDXL graph
evaluated in float64?  Separates kernel error from the fp32 conditioning of a 34-transformer-block random-weight network
(tests/test_gpu_fullsize.py::test_sdxl_1024px_two_taps quotes the result).  Test infrastructure (it runs the oracle), not
collected by pytest.  Usage: python tests/probe_sdxl_f64.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsim_amd import config as C, synth as S          # noqa: E402
from oracle import cpu_ref as R                          # noqa: E402
from tests.test_gpu_fullsize import _oracle_unet, _qkv_at_taps   # noqa: E402


def main():
    cfg = C.SDXL
    drop = ("up_blocks.1", "up_blocks.2", "conv_norm_out", "conv_out", "up_blocks.0.attentions.2", "up_blocks.0.resnets.2",
            "up_blocks.0.upsamplers")
    shapes = C.unet_param_shapes(cfg)
    sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if not k.startswith(drop)])
    ctx, pooled = S.make_context(cfg), S.make_pooled(cfg)
    g = torch.Generator("cpu").manual_seed(1234)
    shp = (1, 4, 128, 128)
    zA, zB = torch.randn(shp, generator=g), torch.randn(shp, generator=g)
    n = S.draw_pair_noise(2334, shp)
    taps = {"a": ("up_blocks", [0, 0, 0]), "b": ("up_blocks", [0, 1, 9])}
    res = {}
    for name, dt in (("cpu_f32", torch.float32), ("cpu_f64", torch.float64)):
        unet = _oracle_unet(R, R.SDXL, sd, shapes, dt)
        t0 = time.time()
        feats = []
        for z, nz in ((zA, n[2]), (zB, n[3])):
            x, t = R.sdxl_inputs(z, nz, 600)
            added = {"text_embeds": pooled.to(dt), "time_ids": R.sdxl_time_ids(unet.cfg).repeat(2, 1).to(dt)}
            feats.append(_qkv_at_taps(R, unet, torch.cat([x] * 2).to(dt), t, ctx.to(dt), added, taps))
        res[name] = {k: float(R.pair_score(*feats[0][k], *feats[1][k], "cosine")) for k in taps}
        print(name, res[name], f"{time.time() - t0:.0f} s", flush=True)
        del unet
    from diffsim_amd.diffsim_xl import diffsim_xl
    xl = diffsim_xl(torch.float32, "cuda", unet_config=cfg, state_dict=sd)
    res["hip_f32"] = {k: float(xl.score_latent_pairs(zA, zB, n[2], n[3], ctx, pooled, blk, tl, 600, "cosine").cpu())
                      for k, (blk, tl) in taps.items()}
    print("hip_f32", res["hip_f32"])
    for k in taps:
        f64 = res["cpu_f64"][k]
        print(f"tap {k}: |hip_f32 - f64|/f64 = {abs(res['hip_f32'][k] - f64) / abs(f64):.3e}   "
              f"|cpu_f32 - f64|/f64 = {abs(res['cpu_f32'][k] - f64) / abs(f64):.3e}   "
              f"|hip_f32 - cpu_f32|/cpu = {abs(res['hip_f32'][k] - res['cpu_f32'][k]) / abs(res['cpu_f32'][k]):.3e}")


if __name__ == "__main__":
    main()
