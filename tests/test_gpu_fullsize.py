"""Config-size parity (BASELINE.json configs 3, 4, 5 and the 512-px VAE of the pixels-in path), each against the fp32 CPU
oracle on 1 pair: the fp32 kernel mode at the north_star tolerance (score within 1e-4 relative), the bf16 / fp8
production modes under stated bounds.  The tiny-config tests pin the graphs; these pin the kernel shape families that
only appear at full size: SDXL's head_dim 64 at 4096 / 1024 tokens, depth-10 transformers, 2048-wide context; DiT-XL/2's
1152-wide layers and head_dim 72; the VAE's 4096-token 512-d mid attention and 512x512x128 convolutions."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsim_amd import config as C
from diffsim_amd import synth as S

REL_F32 = 1e-4


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-6)


def _oracle_unet(R, rcfg, sd, shapes, dtype=torch.float32):
    """Oracle U-Net without materialising random-init parameters first: meta construction + assign (the unused tail of
    the graph gets zero tensors, which calloc never touches)."""
    with torch.device("meta"):
        m = R.UNet2DConditionModel(rcfg)
    full = {k: (sd[k].to(dtype) if k in sd else torch.zeros(shp, dtype=dtype)) for k, shp in shapes.items()}
    m.load_state_dict(full, strict=True, assign=True)
    return m.eval()


def _qkv_at_taps(R, unet, x, t, ctx, added, taps):
    """q,k,v of several taps from ONE oracle forward (pre-hooks, diffsim/diffsim.py:43-56 style); stops at the last."""
    store, hooks = {}, []
    names = list(taps)
    for name in names:
        mod = unet.tap_module(*taps[name])

        def hook(m, inp, name=name):
            store[name] = m.qkv(inp[0])
            if name == names[-1]:
                raise R._TapReached()
        hooks.append(mod.register_forward_pre_hook(hook))
    try:
        with torch.no_grad():
            unet.forward(x, t, ctx, None, added)
    except R._TapReached:
        pass
    finally:
        for h in hooks:
            h.remove()
    return store


def test_sdxl_1024px_two_taps():
    """Config 4: the SDXL U-Net at 1024 px (128 x 128 latents), taps up_blocks [0,0,0] and [0,1,9] (the last block of
    a depth-10 transformer; diffsim/diffsim_xl.py:88-107), both from one shared weight copy.

    The oracle's U-Net runs in fp32 (the north_star's "reference CPU path in fp32"); only the LAST op of its score tail,
    F.cosine_similarity over 2.6 M elements (diffsim/diffsim.py:187-190), is evaluated in float64.  That one op is the whole
    1.0e-4 distance between the fp32 CPU oracle and the float64 evaluation of this graph (tests/probe_sdxl_f32_ops.py ->
    profiles/r03_sdxl_f32_ops_probe.txt: fp32 features + float64 cosine are 1.5e-8 from float64; with torch's fp32 CPU
    reduction 1.03e-4) -- against the all-fp32 tail the 1e-4 gate would measure that reduction's rounding, not the kernels'.
    (The HIP tail folds its partial sums in float64 in a fixed order; the all-fp32 CPU score is asserted to stay within 3e-4.)"""
    from oracle import cpu_ref as R
    from diffsim_amd.diffsim_xl import diffsim_xl
    cfg = C.SDXL
    drop = ("up_blocks.1", "up_blocks.2", "conv_norm_out", "conv_out", "up_blocks.0.attentions.2", "up_blocks.0.resnets.2",
            "up_blocks.0.upsamplers")
    shapes = C.unet_param_shapes(cfg)
    sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if not k.startswith(drop)])
    f64 = torch.float64
    unet = _oracle_unet(R, R.SDXL, sd, shapes, torch.float32)
    ctx, pooled = S.make_context(cfg), S.make_pooled(cfg)
    g = torch.Generator("cpu").manual_seed(1234)
    shp = (1, 4, 128, 128)
    zA, zB = torch.randn(shp, generator=g), torch.randn(shp, generator=g)
    n = S.draw_pair_noise(2334, shp)
    taps = {"a": ("up_blocks", [0, 0, 0]), "b": ("up_blocks", [0, 1, 9])}
    feats = []
    for z, nz in ((zA, n[2]), (zB, n[3])):
        x, t = R.sdxl_inputs(z, nz, 600)
        added = {"text_embeds": pooled, "time_ids": R.sdxl_time_ids(unet.cfg).repeat(2, 1)}
        feats.append(_qkv_at_taps(R, unet, torch.cat([x] * 2), t, ctx, added, taps))
    want = {name: float(R.pair_score(*[f.to(f64) for f in feats[0][name]], *[f.to(f64) for f in feats[1][name]], "cosine")) for name in taps}
    want32 = {name: float(R.pair_score(*feats[0][name], *feats[1][name], "cosine")) for name in taps}
    del unet
    xl = diffsim_xl(torch.float32, "cuda", unet_config=cfg, state_dict=sd)
    for name, (blk, tl) in taps.items():
        s = float(xl.score_latent_pairs(zA, zB, n[2], n[3], ctx, pooled, blk, tl, 600, "cosine").cpu())
        assert _rel(s, want[name]) <= REL_F32, (name, s, want[name])
        assert _rel(s, want32[name]) <= 3e-4, (name, s, want32[name])       # the all-fp32 CPU tail: its own reduction error
    assert xl._base is not None and len(xl._engines) == 2          # two taps, ONE packed weight copy
    q, k, v = xl.features(zB, n[3], ctx, pooled, "up_blocks", [0, 0, 0], 600)
    for got, ref in zip((q, k, v), feats[1]["a"]):
        w = ref.transpose(1, 2).reshape(2, ref.shape[2], -1).float()
        assert (got[0].float().cpu() - w).abs().max().item() <= 2e-4 * float(w.abs().max())
    del xl
    torch.cuda.empty_cache()
    xb = diffsim_xl(torch.bfloat16, "cuda", unet_config=cfg, state_dict=sd)
    for name, (blk, tl) in taps.items():
        s = float(xb.score_latent_pairs(zA, zB, n[2], n[3], ctx, pooled, blk, tl, 600, "cosine").cpu())
        assert abs(s - want[name]) <= 1e-2, (name, s, want[name])          # bf16 production mode: absolute bound


def test_dit_xl2_256px_fp32_bf16_fp8():
    """Config 5: DiT-XL/2 (1152 wide, 16 heads x 72, 256 tokens) at 256 px, tap blocks[13], step 600."""
    from oracle import cpu_ref as R
    from diffsim_amd.diffsim_dit import diffsim_DiT
    cfg = C.DIT_XL2
    keys = [k for k in C.dit_param_shapes(cfg) if not (k.startswith("blocks.") and int(k.split(".")[1]) > 13)]
    sd = S.make_state_dict(cfg, seed=0, keys=keys)
    with torch.device("meta"):
        m = R.DiTOracle(R.DIT_XL2)
    full = {k: (sd[k] if k in sd else torch.zeros(v.shape)) for k, v in m.state_dict().items()}
    m.load_state_dict(full, strict=True, assign=True)
    m.eval()
    g = torch.Generator("cpu").manual_seed(1234)
    shp = (1, 4, 32, 32)
    zA, zB = torch.randn(shp, generator=g), torch.randn(shp, generator=g)
    n = S.draw_pair_noise(2334, shp)
    want = float(R.diffsim_dit_latents(m, zA, zB, n[2], n[3], 600, 13, "cosine"))
    s32 = float(diffsim_DiT(256, 600, "cuda", dit_config=cfg, state_dict=sd, torch_dtype=torch.float32)
                .score_latent_pairs(zA, zB, n[2], n[3], 13, 600, "cosine").cpu())
    assert _rel(s32, want) <= REL_F32, (s32, want)
    s16 = float(diffsim_DiT(256, 600, "cuda", dit_config=cfg, state_dict=sd, torch_dtype=torch.bfloat16)
                .score_latent_pairs(zA, zB, n[2], n[3], 13, 600, "cosine").cpu())
    assert abs(s16 - want) <= 1e-2, (s16, want)
    s8 = float(diffsim_DiT(256, 600, "cuda", dit_config=cfg, state_dict=sd, torch_dtype=torch.bfloat16, fp8_attention=True)
               .score_latent_pairs(zA, zB, n[2], n[3], 13, 600, "cosine").cpu())
    assert abs(s8 - want) <= 2e-2 and s8 != s16, (s8, s16, want)      # fp8 (e4m3) attention: opt-in, looser stated bound


def test_vae_sd15_512px():
    """The SD1.5 VAE encoder at 512 px (diffsim/diffsim.py:92-96): 4096-token single-head 512-d mid attention,
    512x512x128 convolutions."""
    from oracle import cpu_ref as R
    from diffsim_amd.engine import VAEEncoder
    cfg = C.VAE_SD15
    sd = S.make_state_dict(cfg, seed=3)
    ref = R.AutoencoderKLEncoder(R.VAE_SD15)
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    ref.eval()
    a, b = S.make_image_pair(0, 512)
    with torch.no_grad():
        want = ref.moments(a)
    enc32 = VAEEncoder(cfg, sd, torch.float32)
    got = enc32.moments(a).cpu()
    assert got.shape == want.shape == (1, 8, 64, 64)
    err = (got - want).abs().max().item()
    assert err <= 2e-4 * max(float(want.abs().max()), 1.0), err
    # two images: the mid attention's q k^T and P v run as ONE launch each with per-image weights (GemmArgs.wb_rows)
    got2 = enc32.moments(torch.cat([b, a])).cpu()
    err2 = (got2[1:] - want).abs().max().item()
    assert err2 <= 2e-4 * max(float(want.abs().max()), 1.0), err2
    gotb = VAEEncoder(cfg, sd, torch.bfloat16).moments(a).cpu()
    errb = (gotb - want).abs().max().item()
    assert errb <= 6e-2 * max(float(want.abs().max()), 1.0), errb
    # mean error of the bf16 production mode relative to the mean magnitude of the moments (what feeds the latents):
    # measured 1.1 %, bound 2 %
    assert float((gotb - want).abs().mean()) <= 2e-2 * float(want.abs().mean() + 1e-6)


def test_nights_shaped_triplets_full_size():
    """Config 3 shape at full size: SD1.5 512-px triplets (ref, left, right) scored with the cached reference image
    (3 forwards per triplet) -- fp32 scores against per-pair oracle calls at 1e-4, bf16 decisions equal to the oracle's."""
    from oracle import cpu_ref as R
    from diffsim_amd import harness as H
    from diffsim_amd.diffsim import DiffSim
    cfg = C.SD15
    shapes = C.unet_param_shapes(cfg)
    sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if not k.startswith(("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out"))])
    unet = _oracle_unet(R, R.SD15, sd, shapes)
    ctx = S.make_context(cfg)
    n = S.draw_pair_noise(2334, (1, 4, 64, 64))
    lat = [S.make_pair_latents(cfg, i) for i in range(3)]
    ref = torch.cat([lat[0][0], lat[1][0]]); left = torch.cat([lat[0][1], lat[1][1]]); right = torch.cat([lat[2][0], lat[2][1]])
    fr = [R.features(unet, ref[i:i + 1], n[2], ctx) for i in range(2)]
    fl = [R.features(unet, left[i:i + 1], n[3], ctx) for i in range(2)]
    fg = [R.features(unet, right[i:i + 1], n[3], ctx) for i in range(2)]
    want_l = [float(R.pair_score(*fr[i], *fl[i])) for i in range(2)]
    want_r = [float(R.pair_score(*fr[i], *fg[i])) for i in range(2)]
    del unet
    ds = DiffSim(torch_dtype=torch.float32, device="cuda", unet_config=cfg, state_dict=sd)
    sl, sr, bad = H.score_latent_triplets(ds, ref, left, right, n[2], n[3], ctx, return_status=True)
    assert int(bad) == 0
    for got, want in zip(sl.tolist() + sr.tolist(), want_l + want_r):
        assert _rel(got, want) <= REL_F32, (got, want)
    del ds
    torch.cuda.empty_cache()
    db = DiffSim(torch_dtype=torch.bfloat16, device="cuda", unet_config=cfg, state_dict=sd)
    bl, br = H.score_latent_triplets(db, ref, left, right, n[2], n[3], ctx)
    for got, want in zip(bl.tolist() + br.tolist(), want_l + want_r):
        assert abs(got - want) <= 5e-3, (got, want)
    assert H.nights_decisions(bl.cpu(), br.cpu(), "cosine").tolist() == \
        H.nights_decisions(torch.tensor(want_l), torch.tensor(want_r), "cosine").tolist()


def test_sd15_full_size_ragged_side_28():
    """--image_size 224: the FULL-size SD1.5 graph at latent side 28 (28 -> 14 -> 7 -> 4), where the kernel families the tiny
    config never reaches run on odd maps: ff_fused / rowlin with M % 128 != 0, one-pass GroupNorm at HW = 49 and 16 with
    C = 1280, attn_short / attn_kernel with ragged query counts (784, 196, 49), gemm_skinny and 128 x 80 convs on
    non-power-of-two maps, the CONV3 instantiation instead of CONV3P, resize_nearest at C = 1280 (4 -> 7: the tap
    ("up_blocks", 0) sits behind that explicit-size upsample).  fp32 kernel mode against the oracle at 1e-4 for one pair per
    call (small-batch kernels) and inside a batch of 32 pairs (regular tiles); bf16 / fp16 within their bounds, and rows of the
    1-pair call bit-equal to the same pair inside the 32-pair batch."""
    from oracle import cpu_ref as R
    from diffsim_amd.diffsim import DiffSim
    cfg = C.SD15
    shapes = C.unet_param_shapes(cfg)
    sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if not k.startswith(("conv_norm_out", "conv_out"))])
    unet = _oracle_unet(R, R.SD15, sd, shapes)
    ctx = S.make_context(cfg)
    side, npairs = 28, 32
    g = torch.Generator("cpu").manual_seed(2800)
    zA, zB = (0.18215 * 4.0 * torch.randn((npairs, 4, side, side), generator=g) for _ in range(2))
    nA, nB = (torch.randn((1, 4, side, side), generator=g) for _ in range(2))
    taps = (("up_blocks", 0), ("up_blocks", 2), ("down_blocks", 1), ("mid_blocks", 0))
    probe = (0, 31)                                  # pairs checked against the oracle
    want = {(t, i): float(R.diffsim_latents(unet, zA[i:i + 1], zB[i:i + 1], nA, nB, ctx, 600, t[0], t[1], "cosine"))
            for t in taps for i in probe}
    del unet
    ds = DiffSim(torch_dtype=torch.float32, device="cuda", unet_config=cfg, state_dict=sd)
    for t in taps:
        one = float(ds.diffsim_latents(zA[:1], zB[:1], nA, nB, ctx, t[0], t[1], 600, "cosine").cpu())
        assert _rel(one, want[(t, 0)]) <= REL_F32, (t, one, want[(t, 0)])
        allp = ds.score_latent_pairs(zA, zB, nA, nB, ctx, t[0], t[1], 600, "cosine", batch_pairs=npairs).cpu()
        for i in probe:
            assert _rel(float(allp[i]), want[(t, i)]) <= REL_F32, (t, i, float(allp[i]), want[(t, i)])
    eng = ds.engine("up_blocks", 0)
    assert eng.tokens == 7 * 7                       # behind the 4 -> 7 explicit-size upsample
    del ds
    torch.cuda.empty_cache()
    for dtype, bound in ((torch.bfloat16, 5e-3), (torch.float16, 1e-3)):
        d16 = DiffSim(torch_dtype=dtype, device="cuda", unet_config=cfg, state_dict=sd)
        for t in taps:
            allp = d16.score_latent_pairs(zA, zB, nA, nB, ctx, t[0], t[1], 600, "cosine", batch_pairs=npairs).cpu()
            one = d16.score_latent_pairs(zA[:1], zB[:1], nA, nB, ctx, t[0], t[1], 600, "cosine").cpu()
            assert torch.equal(one[0], allp[0]), (dtype, t)          # small-batch kernels == regular tiles, bit for bit
            for i in probe:
                assert abs(float(allp[i]) - want[(t, i)]) <= bound, (dtype, t, i, float(allp[i]), want[(t, i)])
        del d16
        torch.cuda.empty_cache()


def test_sdxl_ragged_side_26():
    """The SDXL graph at latent side 26 (--image_size 208: 26 -> 13 -> 7; the reference accepts any --image_size and keeps its
    native-size time ids): fp32 kernel mode against the oracle (U-Net in fp32, the final cosine in float64 as in
    test_sdxl_1024px_two_taps) at 1e-4 for the depth-10 level at 7 x 7, a tap at 13 x 13 behind the explicit-size 7 -> 13
    upsample and one on the down path, one pair and a batch of 8; bf16 / fp16 within their bounds."""
    from oracle import cpu_ref as R
    from diffsim_amd.diffsim_xl import diffsim_xl
    cfg = C.SDXL
    drop = ("up_blocks.2", "conv_norm_out", "conv_out")
    shapes = C.unet_param_shapes(cfg)
    sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if not k.startswith(drop)])
    unet = _oracle_unet(R, R.SDXL, sd, shapes, torch.float32)
    ctx, pooled = S.make_context(cfg), S.make_pooled(cfg)
    side, npairs = 26, 8
    g = torch.Generator("cpu").manual_seed(2801)
    zA, zB = (torch.randn((npairs, 4, side, side), generator=g) for _ in range(2))
    nA, nB = (torch.randn((1, 4, side, side), generator=g) for _ in range(2))
    taps = (("up_blocks", [0, 0, 0]), ("up_blocks", [1, 2, 1]), ("down_blocks", [1, 1, 0]))
    f64 = torch.float64
    want = {}
    # ONE oracle forward per image serves the three taps (hooks in execution order: the down-path tap, the 7 x 7 level, the 13 x 13
    # level behind the upsample -- what R.features_xl computes per tap, without re-running the graph in front of each)
    order = (2, 0, 1)
    added = {"text_embeds": pooled, "time_ids": R.sdxl_time_ids(unet.cfg).repeat(2, 1)}

    def feats(z, nz):
        x, t = R.sdxl_inputs(z, nz, 600)
        return _qkv_at_taps(R, unet, torch.cat([x] * 2), t, ctx, added, {ti: taps[ti] for ti in order})
    for i in (0, npairs - 1):
        fa, fb = feats(zA[i:i + 1], nA), feats(zB[i:i + 1], nB)
        for ti in range(len(taps)):
            want[(ti, i)] = float(R.pair_score(*[f.to(f64) for f in fa[ti]], *[f.to(f64) for f in fb[ti]], "cosine"))
    del unet
    xl = diffsim_xl(torch.float32, "cuda", unet_config=cfg, state_dict=sd)
    for ti, (blk, tl) in enumerate(taps):
        one = float(xl.score_latent_pairs(zA[:1], zB[:1], nA, nB, ctx, pooled, blk, tl, 600, "cosine").cpu())
        assert _rel(one, want[(ti, 0)]) <= REL_F32, (blk, tl, one, want[(ti, 0)])
        allp = xl.score_latent_pairs(zA, zB, nA, nB, ctx, pooled, blk, tl, 600, "cosine", batch_pairs=npairs).cpu()
        for i in (0, npairs - 1):
            assert _rel(float(allp[i]), want[(ti, i)]) <= REL_F32, (blk, tl, i, float(allp[i]), want[(ti, i)])
    del xl
    torch.cuda.empty_cache()
    for dtype, bound in ((torch.bfloat16, 1e-2), (torch.float16, 2e-3)):
        x16 = diffsim_xl(dtype, "cuda", unet_config=cfg, state_dict=sd)
        for ti, (blk, tl) in enumerate(taps):
            allp = x16.score_latent_pairs(zA, zB, nA, nB, ctx, pooled, blk, tl, 600, "cosine", batch_pairs=npairs).cpu()
            for i in (0, npairs - 1):
                assert abs(float(allp[i]) - want[(ti, i)]) <= bound, (dtype, blk, tl, i, float(allp[i]), want[(ti, i)])
        del x16
        torch.cuda.empty_cache()


def test_fp32_tap_at_the_320_channel_level_with_many_rows():
    """Regression (round 5): the one-launch tapped q | k | v projection (GemmArgs.out_split) needs a tile width that divides the
    split.  In the f32 parity mode an N % 320 == 0 problem with >= 256 big tiles (3 or more images at 64 x 64, taps
    ("down_blocks", 0) / ("up_blocks", 2)) used to fall back to 128-column tiles, whose third tile straddles the q | k boundary at
    column 320 and wrote part of k into q's tensor; the full-size fp32 tests only tapped the 1280-channel level.  Two pairs
    (8 U-Net batch elements, 32768 rows) against the oracle at 1e-4, and the same pairs one per call."""
    from oracle import cpu_ref as R
    from diffsim_amd.diffsim import DiffSim
    cfg = C.SD15
    shapes = C.unet_param_shapes(cfg)
    sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if k.startswith(("conv_in", "time_embedding", "down_blocks.0"))])
    unet = _oracle_unet(R, R.SD15, sd, shapes)
    ctx = S.make_context(cfg)
    lat = [S.make_pair_latents(cfg, i) for i in range(2)]
    zA, zB = torch.cat([p[0] for p in lat]), torch.cat([p[1] for p in lat])
    n = S.draw_pair_noise(2334, lat[0][0].shape)
    want = [float(R.diffsim_latents(unet, zA[i:i + 1], zB[i:i + 1], n[2], n[3], ctx, 600, "down_blocks", 0, "cosine")) for i in range(2)]
    del unet
    ds = DiffSim(torch_dtype=torch.float32, device="cuda", unet_config=cfg, state_dict=sd)
    both = ds.score_latent_pairs(zA, zB, n[2], n[3], ctx, "down_blocks", 0, 600, "cosine", batch_pairs=2).cpu()
    for i in range(2):
        assert _rel(float(both[i]), want[i]) <= REL_F32, (i, float(both[i]), want[i])
        one = ds.score_latent_pairs(zA[i:i + 1], zB[i:i + 1], n[2], n[3], ctx, "down_blocks", 0, 600, "cosine").cpu()
        assert _rel(float(one[0]), want[i]) <= REL_F32, (i, float(one[0]), want[i])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_vae_512px_epilogue_statistics_are_batch_invariant(dtype):
    """The 512 x 512 and 256 x 256 levels of the VAE take their GroupNorm statistics from the producing conv's epilogue
    (GemmArgs.gn_part: per-tile partials in a fixed order, folded in f64).  One image alone fills those levels' 256-row tiles, so
    the tiles -- and with them every partial sum -- are the same at every batch size: the moments of an image are the same bits
    alone, first in a batch of three and last in a batch of three; and they stay within the 16-bit bound of the oracle."""
    from oracle import cpu_ref as R
    from diffsim_amd.engine import VAEEncoder
    cfg = C.VAE_SD15
    sd = S.make_state_dict(cfg, seed=3)
    a, b = S.make_image_pair(0, 512)
    c, _ = S.make_image_pair(1, 512)
    enc = VAEEncoder(cfg, sd, dtype)
    one = enc.moments(a)
    three = enc.moments(torch.cat([a, b, c]))
    assert torch.equal(one[0], three[0])
    assert torch.equal(enc.moments(torch.cat([c, b, a]))[2], one[0])
    assert torch.equal(enc.moments(b)[0], three[1])
    ref = R.AutoencoderKLEncoder(R.VAE_SD15)
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    ref.eval()
    with torch.no_grad():
        want = ref.moments(a)
    err = (one.cpu() - want).abs().max().item()
    assert err <= (6e-2 if dtype == torch.bfloat16 else 1.5e-2) * max(float(want.abs().max()), 1.0), err
