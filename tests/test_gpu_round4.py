"""Round-4 features through the C ABI: the fp16 compute mode (the arithmetic type every reference driver constructs its
pipeline in, /root/reference/cute_main.py:31, /root/reference/diffsim/diffsim.py:82) against the CPU oracle and the fp32
kernel mode, its overflow guard, and bit-identity properties it must share with the bf16 mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsim_amd import config as C
from diffsim_amd import synth as S


def _scorer(cfg, sd, dtype, **kw):
    from diffsim_amd.diffsim import DiffSim
    return DiffSim(torch_dtype=dtype, device="cuda", unet_config=cfg, state_dict=sd, **kw)


@pytest.fixture(scope="module")
def tiny_env():
    from oracle import cpu_ref as R
    sd = S.make_state_dict(C.TINY, seed=0)
    return dict(sd=sd, oracle=R.build_unet(R.TINY, sd), ctx=S.make_context(C.TINY), R=R)


@pytest.mark.parametrize("block,layer,step", [("up_blocks", 0, 600), ("up_blocks", 2, 900), ("down_blocks", 1, 600), ("mid_blocks", 0, 900)])
def test_tiny_fp16_features_and_scores(tiny_env, block, layer, step):
    """torch.float16 is honoured (no silent bf16 substitution): q/k/v come back as fp16 tensors, within an fp16-sized bound of
    the oracle's, tighter than the bf16 mode's; cosine and mse scores within 2e-3 of the oracle."""
    R, unet, ctx = tiny_env["R"], tiny_env["oracle"], tiny_env["ctx"]
    ds = _scorer(C.TINY, tiny_env["sd"], torch.float16)
    assert ds.dtype == torch.float16
    zA, zB = S.make_pair_latents(C.TINY, 3)
    n = S.draw_pair_noise(2334, zA.shape)
    q, k, v = ds.features(torch.cat([zA, zB]), torch.cat([n[2], n[3]]), ctx, block, layer, step)
    assert q.dtype == torch.float16 and k.dtype == torch.float16 and v.dtype == torch.float16
    db = _scorer(C.TINY, tiny_env["sd"], torch.bfloat16)
    qb, kb, vb = db.features(torch.cat([zA, zB]), torch.cat([n[2], n[3]]), ctx, block, layer, step)
    e16 = eb = 0.0
    for img, (z, nz) in enumerate(((zA, n[2]), (zB, n[3]))):
        qo, ko, vo = R.features(unet, z, nz, ctx, step, block, layer)
        for got, gotb, want in ((q, qb, qo), (k, kb, ko), (v, vb, vo)):
            want = want.transpose(1, 2).reshape(2, want.shape[2], -1)
            sc = max(float(want.abs().max()), 1.0)
            e16 = max(e16, (got[img].float().cpu() - want).abs().max().item() / sc)
            eb = max(eb, (gotb[img].float().cpu() - want).abs().max().item() / sc)
    assert e16 <= 6e-3, e16
    assert e16 < eb, (e16, eb)               # three more significant bits than bf16
    for sim in ("cosine", "mse"):
        s = float(ds.diffsim_latents(zA, zB, n[2], n[3], ctx, block, layer, step, sim).cpu())
        so = float(R.diffsim_latents(unet, zA, zB, n[2], n[3], ctx, step, block, layer, sim))
        assert abs(s - so) <= 2e-3 * max(abs(so), 1.0), (sim, s, so)


def test_fp16_batch_invariance_and_fused_kernels(tiny_env):
    """batch-of-N == N singles bit for bit, repeat calls bit-identical, and the multi-operator launches (fusion mask) against
    their unfused chains, in fp16 as in bf16."""
    ds = _scorer(C.TINY, tiny_env["sd"], torch.float16)
    ctx = tiny_env["ctx"]
    lats = [S.make_pair_latents(C.TINY, i) for i in range(5)]
    n = S.draw_pair_noise(2334, lats[0][0].shape)
    zA = torch.cat([p[0] for p in lats]); zB = torch.cat([p[1] for p in lats])
    s_all = ds.score_latent_pairs(zA, zB, n[2], n[3], ctx)
    assert torch.equal(s_all, ds.score_latent_pairs(zA, zB, n[2], n[3], ctx))
    for i in range(5):
        assert torch.equal(ds.score_latent_pairs(zA[i:i + 1], zB[i:i + 1], n[2], n[3], ctx)[0], s_all[i])
    unf = _scorer(C.TINY, tiny_env["sd"], torch.float16, fusion=0)
    s_unf = unf.score_latent_pairs(zA, zB, n[2], n[3], ctx)
    assert (s_all - s_unf).abs().max().item() <= 2e-3


def test_sd15_full_size_fp16_pairs():
    """The real SD1.5 graph at 512 px in fp16 against the fp32 kernel mode (itself within 1e-4 of the CPU oracle:
    test_gpu_e2e.py::test_sd15_full_size_fp32_and_bf16_pairs): score error <= 5e-4 and below the bf16 mode's; every kernel
    family of the step runs its fp16 twin (ff_fused, ln_linear, the pipelined 4096-key attention), no pair flagged."""
    cfg = C.SD15
    keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out"))]
    sd = S.make_state_dict(cfg, seed=0, keys=keys)
    ctx = S.make_context(cfg)
    n = S.draw_pair_noise(2334, (1, 4, 64, 64))
    lats = [S.make_pair_latents(cfg, i) for i in range(4)]
    zA = torch.cat([p[0] for p in lats]); zB = torch.cat([p[1] for p in lats])
    want = _scorer(cfg, sd, torch.float32).score_latent_pairs(zA, zB, n[2], n[3], ctx).cpu()
    s16 = _scorer(cfg, sd, torch.float16)
    got16 = s16.score_latent_pairs(zA, zB, n[2], n[3], ctx)
    gotb = _scorer(cfg, sd, torch.bfloat16).score_latent_pairs(zA, zB, n[2], n[3], ctx).cpu()
    e16 = (got16.cpu() - want).abs().max().item()
    eb = (gotb - want).abs().max().item()
    assert torch.isfinite(got16).all()
    assert e16 <= 5e-4, (e16, eb)
    assert e16 < eb, (e16, eb)
    eng = s16.engine("up_blocks", 0)
    eng.profile(True)
    s16.score_latent_pairs(zA, zB, n[2], n[3], ctx)
    fams = {r[0] for r in eng.profile_records()}
    eng.profile(False)
    for f in ("ff_fused_f16", "ln_linear_f16", "attention_f16_d40_long", "gemm_f16_256x320_conv3p"):
        assert f in fams, (f, sorted(fams))


def test_fp16_overflow_is_reported_per_pair():
    """fp16 tops out at 65504: a pair whose features overflow is flagged by the per-pair status (dsim_pair_score_status) and its
    score is not a silent finite number; its neighbours in the batch are untouched."""
    from diffsim_amd.engine import pair_score
    g = torch.Generator("cpu").manual_seed(0)
    q, k, v = (torch.randn((4, 2, 64, 4 * 32), generator=g).cuda().to(torch.float16) for _ in range(3))
    v[2, 1, 5, 7] = float("inf")                      # what an overflowed activation looks like in fp16
    ia = torch.tensor([0, 0, 2, 1], dtype=torch.int32, device="cuda")
    ib = torch.tensor([1, 2, 3, 3], dtype=torch.int32, device="cuda")
    s, st = pair_score(q, k, v, ia, ib, 4, "cosine", return_status=True)
    assert st.tolist() == [0, 1, 1, 0], st.tolist()
    assert torch.isfinite(s[[0, 3]]).all() and not torch.isfinite(s[[1, 2]]).any()
    # ... and against float64 SDPA on the clean pairs
    qf, kf, vf = (t.double().cpu().view(4, 2, 64, 4, 32).permute(0, 1, 3, 2, 4) for t in (q, k, v))
    import torch.nn.functional as F
    for p, (a, b) in ((0, (0, 1)), (3, (1, 3))):
        oab = F.scaled_dot_product_attention(qf[a], kf[b], vf[b]); oaa = F.scaled_dot_product_attention(qf[a], kf[a], vf[a])
        oba = F.scaled_dot_product_attention(qf[b], kf[a], vf[a]); obb = F.scaled_dot_product_attention(qf[b], kf[b], vf[b])
        w = 0.5 * (F.cosine_similarity(oab.flatten(), oaa.flatten(), dim=0) + F.cosine_similarity(oba.flatten(), obb.flatten(), dim=0))
        assert abs(float(s[p]) - float(w)) <= 2e-3, (p, float(s[p]), float(w))


def test_sd15_full_size_checkpoint_like_statistics():
    """The synthetic weights everywhere else have tame statistics (logit std ~2, no outlier channels).  Here the full-size SD1.5
    graph runs with what trained checkpoints have: outlier channels (x40) out of every resnet conv, one head per
    self-attention with logits x25 (near one-hot rows; at the 4096-key level scores far beyond key tile 0's maximum, so the
    pipelined kernel's exact fallback runs INSIDE the U-Net) and heavy-tailed latents.  fp32 kernel mode against the fp32 CPU
    oracle at the north_star tolerance; the 16-bit modes finite and within a stated bound of it."""
    from oracle import cpu_ref as R
    cfg = C.SD15
    keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out"))]
    sd = S.add_checkpoint_like_outliers(S.make_state_dict(cfg, seed=0, keys=keys))
    full = dict(sd)
    for k, shp in C.unet_param_shapes(cfg).items():
        if k not in full:
            full[k] = torch.zeros(shp)
    unet = R.build_unet(R.SD15, full)
    del full
    ctx = S.make_context(cfg)
    n = S.draw_pair_noise(2334, (1, 4, 64, 64))
    zA, zB = S.make_heavy_tailed_latents(cfg, 0)
    assert float(zA.abs().max()) > 6.0                       # the tails are there
    want = float(R.diffsim_latents(unet, zA, zB, n[2], n[3], ctx))
    s32 = _scorer(cfg, sd, torch.float32)
    got32 = float(s32.score_latent_pairs(zA, zB, n[2], n[3], ctx).cpu())
    assert abs(got32 - want) <= 1e-4 * max(abs(want), 1e-6), (got32, want)
    # the stress is real: in the first 4096-key self-attention the scaled head has rows whose late keys score > 120 log2 units
    # above everything in key tile 0 (2^120 overflows the fast form's bf16 / fp16 P: the exact fallback must run), while the
    # other heads keep the tame statistics
    import math
    q, k, _v = s32.features(torch.cat([zA, zB]), torch.cat([n[2], n[3]]), ctx, "down_blocks", 0, 600)
    sx = (q[0, 1, :, :40].float() @ k[0, 1, :, :40].float().T) * (1.4426950408889634 / math.sqrt(40))
    excess = sx[:, 64:].max(1).values - sx[:, :64].max(1).values
    assert int((excess > 120).sum()) >= 100 and float(sx.std()) > 30.0
    assert float(((q[0, 1, :, 40:80].float() @ k[0, 1, :, 40:80].float().T) / math.sqrt(40)).std()) < 4.0
    del s32, q, k, _v
    for dtype, bound in ((torch.bfloat16, 1e-2), (torch.float16, 2e-3)):        # measured 8e-4 / 2e-4
        got = _scorer(cfg, sd, dtype).score_latent_pairs(zA, zB, n[2], n[3], ctx).cpu()
        assert torch.isfinite(got).all(), dtype
        assert abs(float(got) - want) <= bound, (dtype, float(got), want)


@pytest.mark.parametrize("side", [14, 9, 28])
def test_ragged_latent_sides_match_the_oracle(tiny_env, side):
    """Latent sides that are not a multiple of 2**(levels-1) -- the reference accepts any --image_size (argprocess.py:8;
    224 px -> 28 -> 14 -> 7 -> 4): stride-2 convs take ceil(H / 2) rows and every non-final up block upsamples to the size of
    the next skip (nearest, explicit size: diffusers' forward_upsample_size, hacked_modules.py:531-533) instead of by 2.
    fp32 kernel mode against the oracle at the north_star tolerance for taps on the down path, at the odd mid level and on
    the up path behind an explicit-size upsample; the bf16 mode within its bound."""
    R, unet, ctx = tiny_env["R"], tiny_env["oracle"], tiny_env["ctx"]
    g = torch.Generator("cpu").manual_seed(100 + side)
    zA, zB = (torch.randn((1, 4, side, side), generator=g) for _ in range(2))
    nA, nB = (torch.randn((1, 4, side, side), generator=g) for _ in range(2))
    ds = _scorer(C.TINY, tiny_env["sd"], torch.float32)
    db = _scorer(C.TINY, tiny_env["sd"], torch.bfloat16)
    for block, layer in (("down_blocks", 1), ("mid_blocks", 0), ("up_blocks", 0), ("up_blocks", 1), ("up_blocks", 2)):
        so = float(R.diffsim_latents(unet, zA, zB, nA, nB, ctx, 600, block, layer, "cosine"))
        s = float(ds.diffsim_latents(zA, zB, nA, nB, ctx, block, layer, 600, "cosine").cpu())
        assert abs(s - so) <= 1e-4 * max(abs(so), 1e-6), (side, block, layer, s, so)
        sb = float(db.diffsim_latents(zA, zB, nA, nB, ctx, block, layer, 600, "cosine").cpu())
        assert abs(sb - so) <= 3e-2, (side, block, layer, sb, so)
    # token counts follow the ceil-div pyramid
    lv = [side]
    for _ in range(3):
        lv.append((lv[-1] + 1) // 2)
    eng = ds.engine("up_blocks", 1)          # the second attention-bearing up block: resolution level 1
    eng.set_sample_size(side)
    assert eng.tokens == lv[1] ** 2
    # batch-of-N == N singles at a ragged size too
    z2A, z2B = torch.cat([zA, zB]), torch.cat([zB, zA])
    both = db.score_latent_pairs(z2A, z2B, nA, nB, ctx, "up_blocks", 2, 600, "cosine")
    assert torch.equal(both[0], db.score_latent_pairs(zA, zB, nA, nB, ctx, "up_blocks", 2, 600, "cosine")[0])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_small_batch_gemm_is_bit_identical_to_the_large_batch_path(dtype):
    """Problems too small to fill the chip run on 64 x 64 tiles behind an 8-slot LDS ring (csrc/gemm_skinny.hip) instead of
    gemm_kernel's 128-row tiles.  The per-element arithmetic is the same MFMA sequence with the same rounding points, so a row
    computed in a small batch equals, bit for bit, the same row computed inside a large batch (which takes gemm_kernel):
    3x3 convs (plain, + residual, stride 2, folded x2 upsample), linears (plain, + bias, + residual), ragged M and N edges --
    and both agree with fp32 torch on the CPU."""
    import torch.nn.functional as F
    from diffsim_amd import engine as E
    g = torch.Generator().manual_seed(11)
    rt = 2e-2 if dtype == torch.bfloat16 else 3e-3
    # ---- convs: 3 images alone (M = 3 * 64 = 192 rows at 8 x 8) vs the same 3 images first in a batch of 96
    for Cin, Cout, stride, ups, res in ((320, 320, 1, False, False), (640, 1280, 1, False, True), (128, 192, 2, False, False),
                                        (256, 128, 1, True, True), (1280, 1280, 1, False, True)):
        H = 8
        xs = torch.randn(96, H, H, Cin, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
        b = torch.randn(Cout, generator=g)
        Ho = 2 * H if ups else (H // 2 if stride == 2 else H)
        r = torch.randn(96, Ho, Ho, Cout, generator=g) if res else None
        xd, wd, bd = xs.cuda().to(dtype), w.cuda(), b.cuda()
        rd = r.cuda().to(dtype) if res else None
        small = E.op_conv3x3(xd[:3].contiguous(), wd, bd, rd[:3].contiguous() if res else None, stride, ups)
        big = E.op_conv3x3(xd, wd, bd, rd, stride, ups)
        assert torch.equal(small, big[:3]), (Cin, Cout, stride, ups, res)
        xin = xd[:3].float().cpu().permute(0, 3, 1, 2)
        if ups:
            xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
        want = F.conv2d(xin, w, b, stride=stride, padding=1).permute(0, 2, 3, 1)
        if res:
            want = want + rd[:3].float().cpu()
        assert (small.float().cpu() - want).abs().max().item() <= rt * float(want.abs().max())
    # ---- linears: 200 rows alone vs the same rows first among 40000
    for K, N, res, bias in ((1280, 1280, True, True), (640, 1920, False, False), (2560, 640, True, True), (768, 2560, False, True),
                            (1280, 328, False, True)):
        x = torch.randn(40000, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g) if bias else None
        r = torch.randn(40000, N, generator=g) if res else None
        xd, wd = x.cuda().to(dtype), w.cuda()
        bd = b.cuda() if bias else None
        rd = r.cuda().to(dtype) if res else None
        small = E.op_linear(xd[:200].contiguous(), wd, bd, rd[:200].contiguous() if res else None)
        big = E.op_linear(xd, wd, bd, rd)
        assert torch.equal(small, big[:200]), (K, N, res, bias)
        want = xd[:200].float().cpu() @ w.T + (b if bias else 0.0)
        if res:
            want = want + rd[:200].float().cpu()
        assert (small.float().cpu() - want).abs().max().item() <= rt * float(want.abs().max())


def test_tapped_qkv_as_one_launch_is_bit_identical():
    """DSIM_FUSE_TAPQKV: the tapped layer's to_q / to_k / to_v as one N = 3C launch writing three tensors (taken when q, k, v are
    one allocation, as engine.qkv makes them) against the three separate launches: bit for bit, bf16 / fp16 / fp32, at a width
    the 320-column tiles divide (SD1.5's tap, 1280) -- and the fallback to three launches when the caller's q, k, v are
    separate tensors."""
    from diffsim_amd import _lib
    cfg = C.SD15
    keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out"))]
    sd = S.make_state_dict(cfg, seed=0, keys=keys)
    ctx = S.make_context(cfg)
    n = S.draw_pair_noise(2334, (1, 4, 64, 64))
    zA, zB = S.make_pair_latents(cfg, 1)
    lat, nz = torch.cat([zA, zB]), torch.cat([n[2], n[3]])
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        one = _scorer(cfg, sd, dtype)
        three = _scorer(cfg, sd, dtype, fusion=_lib.FUSE_ALL & ~_lib.FUSE_TAPQKV)
        q1, k1, v1 = one.features(lat, nz, ctx, "up_blocks", 0, 600)
        q3, k3, v3 = three.features(lat, nz, ctx, "up_blocks", 0, 600)
        assert q1.data_ptr() + q1.numel() * q1.element_size() == k1.data_ptr()          # one allocation
        assert torch.equal(q1, q3) and torch.equal(k1, k3) and torch.equal(v1, v3), dtype
        # separate output tensors: the engine falls back to three launches, same bits
        eng = one.engine("up_blocks", 0)
        sep = tuple(torch.empty_like(q1) for _ in range(3))
        import diffsim_amd.scheduler as sched
        t = sched.timestep_from_index(600)
        eng.set_timestep(t)
        sa, sb = sched.noise_coefficients(t)
        qs, ks, vs = eng.qkv(lat.cuda(), nz.cuda(), sa, sb, ctx.cuda(), out=sep)
        assert torch.equal(qs, q1) and torch.equal(ks, k1) and torch.equal(vs, v1), dtype
        del one, three


def test_fp16_mode_of_the_sdxl_and_dit_scorers(golden_dir):
    """diffsim_xl / diffsim_DiT in fp16 (the dtype the reference constructs them in, diffsim_xl.py / diffsim_dit.py) against the
    goldens the reference's own code produced and the oracle: closer than the bf16 mode on average; the e4m3 attention option of
    the DiT scorer stays a bf16-mode feature and says so."""
    import ast
    import os
    import numpy as np
    from oracle import cpu_ref as R
    from diffsim_amd import _lib
    from diffsim_amd.diffsim_xl import diffsim_xl
    from diffsim_amd.diffsim_dit import diffsim_DiT
    # ---- SDXL topology
    sd = S.make_state_dict(C.SDXL_TINY, seed=0)
    ctx, pooled = S.make_context(C.SDXL_TINY), S.make_pooled(C.SDXL_TINY)
    g = np.load(os.path.join(golden_dir, "g8_sdxl_tiny.npz"))
    zA, zB, nA, nB = (torch.from_numpy(g[k]) for k in ("latA", "latB", "noiseA", "noiseB"))
    x16 = diffsim_xl(torch.float16, "cuda", unet_config=C.SDXL_TINY, state_dict=sd)
    xbf = diffsim_xl(torch.bfloat16, "cuda", unet_config=C.SDXL_TINY, state_dict=sd)
    assert x16.dtype == torch.float16
    e16 = ebf = 0.0
    for ci in range(6):
        blk, tl, step, sim = (str(x) for x in g[f"case_{ci}"])
        tl, step = ast.literal_eval(tl), int(step)
        want = float(g[f"score_{ci}"][0])
        s16 = float(x16.score_latent_pairs(zA, zB, nA, nB, ctx, pooled, blk, tl, step, sim).cpu())
        sbf = float(xbf.score_latent_pairs(zA, zB, nA, nB, ctx, pooled, blk, tl, step, sim).cpu())
        assert abs(s16 - want) <= 4e-3 * max(abs(want), 1.0), (ci, s16, want)
        e16 += abs(s16 - want); ebf += abs(sbf - want)
    assert e16 < ebf, (e16, ebf)
    # ---- DiT
    sdd = S.make_state_dict(C.DIT_TINY, seed=0)
    m = R.DiTOracle(R.DIT_TINY)
    m.load_state_dict(sdd, strict=True)
    m.eval()
    gd = np.load(os.path.join(golden_dir, "g9_dit_tiny.npz"))
    zA, zB, nA, nB = (torch.from_numpy(gd[k]) for k in ("latA", "latB", "noiseA", "noiseB"))
    d16 = diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sdd, torch_dtype=torch.float16)
    dbf = diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sdd, torch_dtype=torch.bfloat16)
    e16 = ebf = 0.0
    for ci in range(4):
        layer, step, sim = (str(x) for x in gd[f"case_{ci}"])
        layer, step = int(layer), int(step)
        so = float(R.diffsim_dit_latents(m, zA, zB, nA, nB, step, layer, sim))
        s16 = float(d16.score_latent_pairs(zA, zB, nA, nB, layer, step, sim).cpu())
        sbf = float(dbf.score_latent_pairs(zA, zB, nA, nB, layer, step, sim).cpu())
        assert abs(s16 - so) <= 4e-3 * max(abs(so), 1.0), (ci, s16, so)
        e16 += abs(s16 - so); ebf += abs(sbf - so)
    assert e16 < ebf, (e16, ebf)
    with pytest.raises((_lib.DsimError, ValueError)):
        d8 = diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sdd, torch_dtype=torch.float16, fp8_attention=True)
        d8.score_latent_pairs(zA, zB, nA, nB, 2, 600, "cosine")
