"""VAE encoder on the HIP engine vs the oracle's torch restatement (the diffusers leaf
semantics themselves are parity-unpinned: Appendix A item 11), and the pixels-in DiffSim path."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsim_amd import config as C
from diffsim_amd import synth as S


def _oracle_vae(rcfg, sd):
    from oracle import cpu_ref as R
    m = R.AutoencoderKLEncoder(rcfg)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return m.eval()


@pytest.mark.parametrize("cfgname,size", [("tiny", 64), ("tiny", 128), ("sd15", 64)])
def test_vae_moments_fp32_and_bf16(cfgname, size):
    from oracle import cpu_ref as R
    from diffsim_amd.engine import VAEEncoder
    cfg, rcfg = (C.VAE_TINY, R.VAE_TINY) if cfgname == "tiny" else (C.VAE_SD15, R.VAE_SD15)
    sd = S.make_state_dict(cfg, seed=3)
    ref = _oracle_vae(rcfg, sd)
    a, b = S.make_image_pair(0, size)
    x = torch.cat([a, b])
    want = ref.moments(x)
    got = VAEEncoder(cfg, sd, torch.float32).moments(x).cpu()
    assert got.shape == want.shape == (2, 8, size // 8, size // 8)
    err = (got - want).abs().max().item()
    assert err <= 2e-4 * max(float(want.abs().max()), 1.0), err
    gotb = VAEEncoder(cfg, sd, torch.bfloat16).moments(x).cpu()
    errb = (gotb - want).abs().max().item()
    assert errb <= 6e-2 * max(float(want.abs().max()), 1.0), errb


def test_vae_sampling_order_and_pixels_in_score(golden_dir):
    """Path-based DiffSim.diffsim with the HIP VAE: the generator is consumed vaeA, vaeB, noiseA, noiseB
    (diffsim.py:109-113) and the score equals the oracle's U-Net + oracle VAE on the same images."""
    from oracle import cpu_ref as R
    from diffsim_amd.diffsim import DiffSim, get_generator
    from diffsim_amd.engine import VAEEncoder
    from diffsim_amd.image import load_image, process_image
    vsd = S.make_state_dict(C.VAE_TINY, seed=3)
    usd = S.make_state_dict(C.TINY, seed=0)
    ctx = S.make_context(C.TINY)
    vae = VAEEncoder(C.VAE_TINY, vsd, torch.float32)
    ds = DiffSim(torch_dtype=torch.float32, device="cuda", unet_config=C.TINY, state_dict=usd, vae=vae,
                 encode_prompt=lambda p: ctx)
    img_a, img_b = os.path.join(golden_dir, "g1_img_c.png"), os.path.join(golden_dir, "g1_img_d.png")
    s = ds.diffsim(img_a, img_b, 128, "a cat", "up_blocks", [0], 600, seed=2334, similarity="cosine")
    assert s.shape == (1,)
    # oracle: same orchestration with the oracle VAE and U-Net
    ovae, ounet = _oracle_vae(R.VAE_TINY, vsd), R.build_unet(R.TINY, usd)
    g = get_generator(2334, "cpu")
    tA = process_image(load_image(img_a), 128).to(torch.float16)
    tB = process_image(load_image(img_b), 128).to(torch.float16)
    zA = ovae.sample(tA, g) * 0.18215
    zB = ovae.sample(tB, g) * 0.18215
    nA, nB = torch.randn(zA.shape, generator=g), torch.randn(zB.shape, generator=g)
    so = R.diffsim_latents(ounet, zA, zB, nA, nB, ctx)
    assert abs(float(s.cpu()) - float(so)) <= 1e-4 * abs(float(so)), (float(s.cpu()), float(so))


def test_pair_path_equals_two_single_encodes(golden_dir):
    """diffsim() sends both images through ONE VAE encode, casts on the device and decodes on two host threads:
    the latents and the score must equal the reference-shaped sequential path (two prepare_image_latents calls)
    bit for bit."""
    from diffsim_amd.diffsim import DiffSim, get_generator
    from diffsim_amd.engine import VAEEncoder
    from diffsim_amd.image import load_image, process_image
    vsd = S.make_state_dict(C.VAE_TINY, seed=3)
    usd = S.make_state_dict(C.TINY, seed=0)
    ctx = S.make_context(C.TINY)
    for dtype in (torch.float32, torch.bfloat16):
        vae = VAEEncoder(C.VAE_TINY, vsd, dtype)
        ds = DiffSim(torch_dtype=dtype, device="cuda", unet_config=C.TINY, state_dict=usd, vae=vae, encode_prompt=lambda p: ctx)
        img_a, img_b = os.path.join(golden_dir, "g1_img_c.png"), os.path.join(golden_dir, "g1_img_d.png")
        s = ds.diffsim(img_a, img_b, 128, "a cat", "up_blocks", [0], 600, seed=2334, similarity="cosine")
        g = get_generator(2334, "cpu")
        tA, tB = process_image(load_image(img_a), 128), process_image(load_image(img_b), 128)
        zA = ds.prepare_image_latents(tA.to(torch.float16), vae, None, g)          # host-side cast, one image per encode
        zB = ds.prepare_image_latents(tB.to(torch.float16), vae, None, g)
        g2 = get_generator(2334, "cpu")
        pA, pB = ds._pair_latents(tA, tB, g2)
        assert torch.equal(pA, zA) and torch.equal(pB, zB)
        nA, nB = torch.randn(zA.shape, generator=g), torch.randn(zB.shape, generator=g)
        s2 = ds.score_latent_pairs(zA.float(), zB.float(), nA, nB, "a cat", "up_blocks", 0, 600, "cosine")
        assert torch.equal(s, s2)
        # the batched path-based entry point: chunked VAE encodes, same scores as one diffsim() call per pair
        sr = ds.diffsim(img_b, img_a, 128, "a cat", "up_blocks", [0], 600, seed=2334, similarity="cosine")
        sp = ds.score_pairs([(img_a, img_b), (img_b, img_a), (img_a, img_b)], 128, "a cat", "up_blocks", [0], 600, seed=2334,
                            batch_pairs=2)
        assert torch.equal(sp[0:1], s) and torch.equal(sp[1:2], sr) and torch.equal(sp[2:3], s)
