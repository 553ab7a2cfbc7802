#!/usr/bin/env python3
"""Pin the [EXT] leaves of oracle/cpu_ref.py against the REAL third-party code the reference calls -- to be run ELSEWHERE.

oracle/cpu_ref.py restates, from published semantics, the parts of the DiffSim path whose arithmetic lives in un-vendored
packages (/root/reference/requirements.txt: diffusers==0.29.2 UNet2DConditionModel / AutoencoderKL / PNDMScheduler; call sites
/root/reference/diffsim/diffsim.py:82,92-96 and diffsim/diffsim_pipeline.py:153-157,177-183,213-221).  Neither diffusers nor
a Stable Diffusion 1.5 checkpoint exists in the build container or on the GPU boxes (no network), so those leaves are marked
"parity unpinned".  This script closes that hole for anyone who HAS both:

    pip install diffusers==0.29.2 transformers==4.44.0 safetensors
    python tools/pin_with_real_diffusers.py --model_path /path/to/stable-diffusion-v1-5 [--pairs 2] [--image_size 512]

It loads the real diffusers modules and the oracle from the SAME checkpoint tensors, feeds both the synthetic inputs the
repo's tests use (diffsim_amd/synth.py: seeded latents, images, the reference's generator draw order), and compares
  * the PNDM table: set_timesteps(1000) -> timesteps[600], alphas_cumprod, add_noise               (exact / 1e-6)
  * VAE moments: AutoencoderKL.encode(x).latent_dist.parameters vs oracle AutoencoderKLEncoder.moments  (<= 2e-4 of the range)
  * q, k, v of the hooked attn1 for every (target_block, target_layer) tap, captured from the real U-Net with a forward
    pre-hook the way /root/reference/diffsim/diffsim.py:43-56 does, vs oracle features()                  (<= 2e-4 of the range)
  * the pair score (cosine and mse) of oracle.pair_score on both feature sets                              (<= 1e-4 relative)
and prints one JSON object; exit status 1 if any bound is exceeded.  Nothing of the reference is copied or shipped: the real
packages are imported where they are installed.  It cannot run in this repository's containers; `oracle/__init__.py` says so.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", required=True, help="diffusers-layout SD1.5 directory (unet/, vae/, scheduler/)")
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--image_size", type=int, default=512)
    ap.add_argument("--target_step", type=int, default=600)
    a = ap.parse_args()
    try:
        from diffusers import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
    except ImportError:
        print("this script needs the real diffusers package (pip install diffusers==0.29.2); see its docstring", file=sys.stderr)
        return 2
    from diffsim_amd import config as C, synth as S
    from oracle import cpu_ref as R

    torch.manual_seed(0)
    out, bad = {}, []

    def check(name, err, bound):
        out[name] = {"err": float(err), "bound": bound}
        if not float(err) <= bound:
            bad.append(name)

    # ---- scheduler table ------------------------------------------------------------------
    sch = PNDMScheduler.from_pretrained(os.path.join(a.model_path, "scheduler"))
    sch.set_timesteps(1000)
    ts = [int(t) for t in sch.timesteps]
    mine = [int(t) for t in R.pndm_timesteps(1000)]
    check("pndm_timesteps_differ", sum(x != y for x, y in zip(ts, mine)) + abs(len(ts) - len(mine)), 0)
    check("alphas_cumprod_max_abs", (sch.alphas_cumprod.float() - R.alphas_cumprod()).abs().max(), 1e-6)
    t = R.timestep_from_index(a.target_step)
    z = torch.randn(1, 4, 8, 8)
    nz = torch.randn(1, 4, 8, 8)
    check("add_noise_max_abs", (sch.add_noise(z, nz, torch.tensor([t])) - R.add_noise(z, nz, t)).abs().max(), 1e-6)

    # ---- VAE encoder ------------------------------------------------------------------------------------------------------
    vae = AutoencoderKL.from_pretrained(os.path.join(a.model_path, "vae")).float().eval()
    vsd = {k: v for k, v in vae.state_dict().items() if k.startswith(("encoder.", "quant_conv."))}
    ovae = R.AutoencoderKLEncoder(R.VAE_SD15)
    ovae.load_state_dict(vsd, strict=True)
    ovae.eval()
    img, _ = S.make_image_pair(0, a.image_size)
    with torch.no_grad():
        want = vae.encode(img).latent_dist.parameters
        got = ovae.moments(img)
    check("vae_moments_rel_range", (got - want).abs().max() / want.abs().max().clamp_min(1.0), 2e-4)

    # ---- U-Net to the tap: q / k / v and scores -----------------------------------------------------------------------------
    unet = UNet2DConditionModel.from_pretrained(os.path.join(a.model_path, "unet")).float().eval()
    ounet = R.build_unet(R.SD15, unet.state_dict())
    ctx = S.make_context(C.SD15)                     # (2, 77, 768) [uncond, cond]: any context pins the arithmetic
    side = a.image_size // 8

    class Reached(Exception):
        pass

    def real_qkv(block, layer, z0, noise):
        """The reference's hook: a forward pre-hook on the tapped attn1 (diffsim/diffsim.py:122-145 slicing), q/k/v by the module's
        own projections and head split (diffsim/hacked_attn.py:61-77)."""
        if block == "down_blocks":
            attn = unet.down_blocks[:-1][layer].attentions[-1].transformer_blocks[-1].attn1
        elif block == "mid_blocks":
            attn = unet.mid_block.attentions[-1].transformer_blocks[-1].attn1
        else:
            attn = unet.up_blocks[1:][layer].attentions[-1].transformer_blocks[-1].attn1
        store = {}

        def hook(m, args, kwargs=None):
            x = args[0] if args else kwargs["hidden_states"]
            h = m.heads
            sp = lambda y: y.view(y.shape[0], -1, h, y.shape[-1] // h).transpose(1, 2)
            store["qkv"] = (sp(m.to_q(x)), sp(m.to_k(x)), sp(m.to_v(x)))
            raise Reached()
        hd = attn.register_forward_pre_hook(hook, with_kwargs=True)
        try:
            xt = sch.add_noise(z0, noise, torch.tensor([t]))
            with torch.no_grad():
                unet(torch.cat([xt] * 2), t, encoder_hidden_states=ctx)
        except Reached:
            pass
        finally:
            hd.remove()
        return store["qkv"]

    taps = [("up_blocks", 0), ("up_blocks", 1), ("up_blocks", 2), ("mid_blocks", 0), ("down_blocks", 0), ("down_blocks", 1),
            ("down_blocks", 2)]
    for block, layer in taps:
        for i in range(a.pairs):
            zA, zB = S.make_pair_latents(C.SD15, i)
            if side != C.SD15.sample_size:
                g = torch.Generator("cpu").manual_seed(4000 + i)
                zA, zB = (0.18215 * torch.randn(1, 4, side, side, generator=g) for _ in range(2))
            n = S.draw_pair_noise(2334, zA.shape)
            fr = [real_qkv(block, layer, zA, n[2]), real_qkv(block, layer, zB, n[3])]
            fo = [R.features(ounet, zA, n[2], ctx, a.target_step, block, layer), R.features(ounet, zB, n[3], ctx, a.target_step, block, layer)]
            for img_i in range(2):
                for nm, x, y in zip("qkv", fr[img_i], fo[img_i]):
                    check(f"{block}{layer}_pair{i}_img{img_i}_{nm}_rel_range", (x - y).abs().max() / y.abs().max().clamp_min(1e-6), 2e-4)
            for sim in ("cosine", "mse"):
                sr = float(R.pair_score(*fr[0], *fr[1], sim))
                so = float(R.pair_score(*fo[0], *fo[1], sim))
                check(f"{block}{layer}_pair{i}_{sim}_rel", abs(sr - so) / max(abs(sr), 1e-6), 1e-4)
    print(json.dumps({"pinned": not bad, "failed": bad, "checks": out}, indent=1))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
