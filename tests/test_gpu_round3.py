"""Round-3 features through the C ABI: process_image's arithmetic and the VAE posterior sampling on the device (bit-identical to
the host path they replace), DiffSim-XL / DiffSim-DiT through the batched, cached triplet harness (3 forwards per triplet, bit-equal
to their per-pair diffsim_score calls), one DiT weight copy for every --target_layer, the Sref driver end to end."""
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsim_amd import config as C
from diffsim_amd import synth as S

NAMES = ["g1_img_a.png", "g1_img_b.png", "g1_img_c.png", "g1_img_d.png"]


def test_image_preprocess_is_bit_identical_to_process_image(golden_dir):
    """dsim_image_preprocess = process_image after its resize (diffsim/diffsim.py:31-41), against the reference-generated
    golden G1 and against every uint8 value; to_half reproduces the SD1.5 pipeline's fp16 image cast (diffsim.py:93)."""
    from diffsim_amd.engine import image_preprocess
    from diffsim_amd.image import load_image, process_image, resize_u8
    g = np.load(os.path.join(golden_dir, "g1_process_image.npz"))        # the reference's own process_image outputs (make_golden.py)
    for name in NAMES:
        letter = name.split(".")[0].split("_")[-1]
        im = load_image(os.path.join(golden_dir, name))
        for size in (64, 128):
            host = process_image(im, size)
            dev = image_preprocess(resize_u8(im, size).cuda()).cpu()
            assert torch.equal(dev, host), (name, size)
            assert np.array_equal(dev.numpy(), g[f"{letter}_{size}"]), (name, size)
            assert torch.equal(image_preprocess(resize_u8(im, size).cuda(), True).cpu(), host.to(torch.float16).float())
    ramp = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16, 1).expand(1, 16, 16, 3).contiguous()
    want = ((ramp.float().numpy() / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)).transpose(0, 3, 1, 2)
    assert np.array_equal(image_preprocess(ramp.cuda()).cpu().numpy(), want)


def test_latent_sample_is_bit_identical_to_the_distribution_sample():
    """dsim_latent_sample = scaling_factor * DiagonalGaussianDistribution.sample() (prepare_image_latents, diffsim.py:92-96)
    with the caller's draw: against the tensor expression it replaces, strided image selection (triplets), per-image draws,
    logvar clamping and the fp16 rounding of the fp16 pipelines."""
    from diffsim_amd.engine import _LatentDist, latent_sample
    g = torch.Generator().manual_seed(7)
    mom = torch.randn(6, 8, 16, 16, generator=g) * 3.0
    mom[0, 4:, :2] = 50.0; mom[1, 4:, :2] = -60.0           # logvar beyond the clamp
    eps1 = torch.randn(1, 4, 16, 16, generator=g)
    epsn = torch.randn(2, 4, 16, 16, generator=g)
    md = mom.cuda()
    for first, stride, eps in ((0, 1, eps1), (1, 3, eps1), (2, 3, epsn), (0, 2, eps1)):
        sel = mom[first::stride].double()
        mean, lv = sel[:, :4], sel[:, 4:].clamp(-30.0, 20.0)
        want = 0.18215 * (mean + torch.exp(0.5 * lv) * eps.double())
        got = latent_sample(md, eps.cuda(), 0.18215, first, stride)
        assert got.shape == want.shape
        scale = 0.18215 * (mean.abs() + (torch.exp(0.5 * lv) * eps.double()).abs())      # (the sum may cancel: error relative to its terms)
        err = (got.cpu().double() - want).abs() / (scale + 1e-30)
        assert float(err.max()) <= 4e-7, (first, stride, float(err.max()))              # f32 rounding of exp, two products and a sum
        assert torch.equal(latent_sample(md, eps.cuda(), 0.18215, first, stride, True), got.to(torch.float16).float())
        # the distribution object of the per-pair path is the same launch with scale 1, then one multiply: bit-identical
        if eps.shape[0] == 1:
            d = _LatentDist(md[first::stride])
            e = torch.randn(d.mean.shape, generator=torch.Generator().manual_seed(5))
            assert torch.equal(0.18215 * d.sample(generator=torch.Generator().manual_seed(5)),
                               latent_sample(md, e.cuda(), 0.18215, first, stride))


def _copy_images(golden_dir, tmp_path):
    for n in NAMES:
        shutil.copy(os.path.join(golden_dir, n), tmp_path / n)
    return [str(tmp_path / n) for n in NAMES]


def test_xl_and_dit_through_the_cached_triplet_harness(golden_dir, tmp_path):
    """harness.score_path_triplets for diffsim_xl and diffsim_DiT scorers: 3 forwards per triplet, chunked batches, one
    context encode per prompt -- and every score bit-equal to the scorer's own per-pair diffsim_score (the two calls per
    triplet of style_main.py:103-107); one DiT handle serves every target layer bit-identically to a fresh handle."""
    from diffsim_amd import harness as Hn
    from diffsim_amd.diffsim_dit import diffsim_DiT
    from diffsim_amd.diffsim_xl import diffsim_xl
    from diffsim_amd.engine import VAEEncoder
    p = _copy_images(golden_dir, tmp_path)
    trip = [(p[0], p[1], p[2], "a cat"), (p[1], p[2], p[3], "a dog"), (p[3], p[0], p[1], "a cat"), (p[2], p[3], p[0], "a cat"),
            (p[0], p[2], p[3], "a dog")]
    vae_sd = S.make_state_dict(C.VAE_TINY, seed=3)
    # ---- DiffSim-XL ----
    xsd = S.make_state_dict(C.SDXL_TINY, seed=0)
    ctx, pooled = S.make_context(C.SDXL_TINY), S.make_pooled(C.SDXL_TINY)
    calls = []

    def enc(prompt):
        calls.append(prompt)
        sc = 1.0 if prompt == "a cat" else 0.5
        return ctx * sc, pooled * sc
    for nd in (torch.float32, torch.float16):
        xl = diffsim_xl(torch.float32, "cuda", unet_config=C.SDXL_TINY, state_dict=xsd,
                        vae=VAEEncoder(C.VAE_TINY, vae_sd, torch.float32), encode_prompt=enc, noise_dtype=nd)
        calls.clear()
        sl, sr, bad = Hn.score_path_triplets(xl, trip, 128, "up_blocks", [0, 1, 2], 600, 2334, "cosine", batch_triplets=2)
        assert bad == 0 and sorted(set(calls)) == ["a cat", "a dog"] and len(calls) == 2        # one encode per prompt
        for j, (a, b, c, prompt) in enumerate(trip):
            ab = xl.diffsim_score(a, b, 128, prompt, "up_blocks", [0, 1, 2], 600, "cosine", 2334)
            ac = xl.diffsim_score(a, c, 128, prompt, "up_blocks", [0, 1, 2], 600, "cosine", 2334)
            assert torch.equal(sl[j:j + 1], ab) and torch.equal(sr[j:j + 1], ac), (nd, j)
    # ---- DiffSim-DiT ----
    dcfg = C.DIT_TINY
    dsd = S.make_state_dict(dcfg, seed=0)
    dit = diffsim_DiT(128, 600, "cuda", dit_config=dcfg, state_dict=dsd, vae=VAEEncoder(C.VAE_TINY, vae_sd, torch.float32),
                      torch_dtype=torch.float32)
    for layer in ([2], [0], [1]):                       # the tap moves on ONE handle (dsim_dit_set_tap)
        sl, sr, bad = Hn.score_path_triplets(dit, trip, 128, "up_blocks", layer, 600, 2334, "mse", batch_triplets=3)
        fresh = diffsim_DiT(128, 600, "cuda", dit_config=dcfg, state_dict=dsd, vae=dit.vae, torch_dtype=torch.float32)
        for j, (a, b, c, prompt) in enumerate(trip):
            ab = fresh.diffsim_score(a, b, 128, prompt, "up_blocks", layer, 600, "mse", 2334)
            ac = fresh.diffsim_score(a, c, 128, prompt, "up_blocks", layer, 600, "mse", 2334)
            assert torch.equal(sl[j:j + 1], ab) and torch.equal(sr[j:j + 1], ac), (layer, j)
    assert dit._engine is not None and dit.engine(1) is dit.engine(2)


def test_sd15_pairs_and_triplets_with_device_preprocessing(golden_dir, tmp_path):
    """DiffSim.score_pairs / score_path_triplets (decode + resize on the host pool two chunks ahead, everything else on the
    device) against per-pair DiffSim.diffsim calls: bit-equal, in both pipeline dtypes."""
    from diffsim_amd import harness as Hn
    from diffsim_amd.diffsim import DiffSim
    from diffsim_amd.engine import VAEEncoder
    p = _copy_images(golden_dir, tmp_path)
    sd = S.make_state_dict(C.TINY, seed=0)
    ctx = S.make_context(C.TINY)
    for nd in (torch.float32, torch.float16):
        ds = DiffSim(torch_dtype=torch.float32, device="cuda", unet_config=C.TINY, state_dict=sd,
                     vae=VAEEncoder(C.VAE_TINY, S.make_state_dict(C.VAE_TINY, seed=3), torch.float32), encode_prompt=lambda q: ctx,
                     noise_dtype=nd)
        pairs = [(p[0], p[1]), (p[2], p[3]), (p[1], p[3]), (p[3], p[0]), (p[2], p[0])]
        got = ds.score_pairs(pairs, 128, "a cat", "up_blocks", [0], 600, seed=2334, similarity="cosine", batch_pairs=2)
        trip = [(a, b, p[(i + 2) % 4], "a cat") for i, (a, b) in enumerate(pairs)]
        sl, sr, _ = Hn.score_path_triplets(ds, trip, 128, "up_blocks", [0], 600, 2334, "cosine", batch_triplets=2)
        for i, (a, b) in enumerate(pairs):
            want = ds.diffsim(a, b, 128, "a cat", "up_blocks", [0], 600, seed=2334, similarity="cosine")
            assert torch.equal(got[i:i + 1], want) and torch.equal(sl[i:i + 1], want), (nd, i)
            assert torch.equal(sr[i:i + 1], ds.diffsim(a, trip[i][2], 128, "a cat", "up_blocks", [0], 600, seed=2334, similarity="cosine"))
