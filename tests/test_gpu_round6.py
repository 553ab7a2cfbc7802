"""Round 6: the persistent score-tail kernel at SD1.5's default tap (csrc/attn160.hip: 256 tokens, 8 heads x 160, 16-bit compute
types) against the reference's tail arithmetic restated in float64 torch (/root/reference/diffsim/diffsim.py:177-197), through the
C ABI (dsim_pair_score / dsim_pair_score_status)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B, H, N, D = 2, 8, 256, 160


@pytest.fixture(scope="module")
def eng():
    from diffsim_amd import engine
    return engine


def _tail_reference(q, k, v, ia, ib, sim, out_dtype):
    """float64 SDPAs on the rounded operands; the SDPA outputs rounded to the pipeline dtype (what torch's SDPA returns in the
    reference's fp16 pipeline), the cosine / mse in float64"""
    def heads(t):
        return t.double().view(B, N, H, D).transpose(1, 2)

    def sdpa(qq, kk, vv):
        return F.scaled_dot_product_attention(heads(qq), heads(kk), heads(vv)).to(out_dtype).double()

    res = []
    for a, b in zip(ia.tolist(), ib.tolist()):
        o_ab, o_aa = sdpa(q[a], k[b], v[b]), sdpa(q[a], k[a], v[a])
        o_ba, o_bb = sdpa(q[b], k[a], v[a]), sdpa(q[b], k[b], v[b])
        if sim == "cosine":
            s = 0.5 * (F.cosine_similarity(o_ab.reshape(1, -1), o_aa.reshape(1, -1)) + F.cosine_similarity(o_ba.reshape(1, -1), o_bb.reshape(1, -1)))
        else:
            s = 0.5 * (F.mse_loss(o_ab, o_aa) + F.mse_loss(o_ba, o_bb))
        res.append(float(s))
    return torch.tensor(res, dtype=torch.float64)


def _features(nf, seed, dtype, logit_scale=1.0, correlate=0.0):
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(nf, B, N, H * D, generator=g) for _ in range(3))
    if correlate:                   # images that resemble image 0: scores away from zero
        for t in (q, k, v):
            t[1:] = correlate * t[:1] + (1 - correlate) * t[1:]
    q = q * logit_scale
    return tuple(t.to(dtype).cuda().contiguous() for t in (q, k, v))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sim", ["cosine", "mse"])
def test_persistent_tail_matches_the_float64_tail(eng, dtype, sim):
    """Five pairs over four images (a shared image, a pair of an image with itself), both similarities, both 16-bit types."""
    q, k, v = _features(4, 11, dtype, correlate=0.6)
    ia = torch.tensor([0, 0, 2, 3, 1], dtype=torch.int32).cuda()
    ib = torch.tensor([1, 2, 2, 0, 3], dtype=torch.int32).cuda()
    got = eng.pair_score(q, k, v, ia, ib, H, sim).double().cpu()
    want = _tail_reference(q.cpu(), k.cpu(), v.cpu(), ia.cpu(), ib.cpu(), sim, dtype)
    # the kernel's P is rounded to 8 (bf16) / 11 (fp16) bits before the PV products; the reference's is not
    tol = 4e-3 if dtype == torch.bfloat16 else 5e-4
    if sim == "cosine":
        assert (got - want).abs().max().item() <= tol, (got, want)
        assert abs(float(got[2]) - 1.0) <= 1e-6          # an image against itself: both passes run the same arithmetic
    else:
        assert ((got - want).abs() / want.abs().clamp_min(1e-6)).max().item() <= 10 * tol, (got, want)
        assert float(got[2]) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_persistent_tail_with_peaked_logits_takes_the_rescale_path(eng, dtype):
    """Logits with a standard deviation of ~25 log2 units: row maxima grow by more than the 8-unit threshold in most steps, so the
    softmax reference point moves and O / l are rescaled (the branch bounded-random data never takes)."""
    q, k, v = _features(3, 12, dtype, logit_scale=14.0, correlate=0.5)
    ia = torch.tensor([0, 1], dtype=torch.int32).cuda()
    ib = torch.tensor([1, 2], dtype=torch.int32).cuda()
    got = eng.pair_score(q, k, v, ia, ib, H, "cosine").double().cpu()
    want = _tail_reference(q.cpu(), k.cpu(), v.cpu(), ia.cpu(), ib.cpu(), "cosine", dtype)
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= (1e-2 if dtype == torch.bfloat16 else 2e-3), (got, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_persistent_tail_is_deterministic_and_batch_invariant(eng, dtype):
    """33 pairs over 6 images: more units than one round of workgroups' first units on small grids, every workgroup path with a next
    unit; repeat calls bit-identical; batch-of-N == N singles bit for bit; swapping the roles of a pair changes nothing."""
    q, k, v = _features(6, 13, dtype, correlate=0.4)
    g = torch.Generator().manual_seed(3)
    ia = torch.randint(0, 6, (33,), generator=g, dtype=torch.int32).cuda()
    ib = torch.randint(0, 6, (33,), generator=g, dtype=torch.int32).cuda()
    s1 = eng.pair_score(q, k, v, ia, ib, H, "cosine")
    s2 = eng.pair_score(q, k, v, ia, ib, H, "cosine")
    assert torch.equal(s1, s2)
    for p in (0, 7, 32):
        sp = eng.pair_score(q, k, v, ia[p:p + 1].clone(), ib[p:p + 1].clone(), H, "cosine")
        assert torch.equal(sp[0], s1[p])
    s8 = eng.pair_score(q, k, v, ia[:8].clone(), ib[:8].clone(), H, "cosine")
    assert torch.equal(s8, s1[:8])
    s3 = eng.pair_score(q, k, v, ib, ia, H, "cosine")
    assert torch.equal(s1, s3)
    m1 = eng.pair_score(q, k, v, ia, ib, H, "mse")
    assert torch.equal(m1, eng.pair_score(q, k, v, ib, ia, H, "mse"))


def test_persistent_tail_reports_non_finite_features_per_pair(eng):
    q, k, v = _features(4, 14, torch.float16)
    v[2, 1, 17, 300] = float("inf")
    ia = torch.tensor([0, 2, 1, 3], dtype=torch.int32).cuda()
    ib = torch.tensor([1, 3, 3, 2], dtype=torch.int32).cuda()
    s, st = eng.pair_score(q, k, v, ia, ib, H, "cosine", return_status=True)
    assert st.cpu().tolist() == [0, 1, 0, 1]
    assert torch.isfinite(s[[0, 2]]).all()


# ---- the same core as the U-Net's own 256-token self-attention (sdpa160_kernel), through dsim_op_attention ------------------------
def _attention_fused_rows(qkv, out_ld=None):
    """qkv: [B][256][3 C] cuda tensor as the fused q|k|v projection writes it (q, k, v are column blocks of one row) -> [B][256][C]"""
    from diffsim_amd import _lib, engine
    L = _lib.lib()
    Bn, Nn, C3 = qkv.shape
    C = C3 // 3
    out = torch.full((Bn, Nn, C), float("nan"), dtype=qkv.dtype, device=qkv.device)
    es = qkv.element_size()
    _lib.check(L.dsim_op_attention(qkv.data_ptr(), C3, qkv.data_ptr() + C * es, qkv.data_ptr() + 2 * C * es, C3, out.data_ptr(), C, Bn, Bn, H,
                                   Nn, Nn, C // H, engine._TORCH2DSIM[qkv.dtype], engine._stream_ptr()), "op_attention")
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nb", [3, 40])
def test_persistent_self_attention_matches_float64_sdpa(eng, dtype, nb):
    """nb = 3: fewer units than workgroups; nb = 40: 320 (batch element, head) units on 256 workgroups -- some walk two units (the
    next unit's Q pieces under the current one, the output stores in flight under the next unit's first steps), most end after one."""
    g = torch.Generator().manual_seed(21 + nb)
    qkv = torch.randn(nb, N, 3 * H * D, generator=g)
    qkv[..., : 2 * H * D] *= 1.3
    qkv = qkv.to(dtype).cuda().contiguous()
    got = _attention_fused_rows(qkv)
    assert torch.isfinite(got.float()).all()
    x = qkv.cpu().double().view(nb, N, 3, H, D)
    want = F.scaled_dot_product_attention(x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2))
    want = want.transpose(1, 2).reshape(nb, N, H * D)
    # P is rounded to the 16-bit type before the PV products (a random error of ~2^-9 / 2^-12 of the dominant |p v| terms, maximum
    # taken over up to 1.3e7 values) and the output once more (half a unit in its last place)
    atol, rtol = (4e-3, 1.6e-2) if dtype == torch.bfloat16 else (5e-4, 2e-3)
    excess = ((got.cpu().double() - want).abs() / (atol + rtol * want.abs())).max().item()
    assert excess <= 1.0, excess
    # repeat calls bit-identical; a batch element's rows do not depend on the batch it is computed in
    assert torch.equal(got, _attention_fused_rows(qkv))
    for b in (0, nb - 1):
        assert torch.equal(got[b], _attention_fused_rows(qkv[b:b + 1].contiguous())[0])


def test_persistent_self_attention_peaked_rows(eng):
    """logits with ~25 log2 units of spread: the rescale branch"""
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(2, N, 3 * H * D, generator=g)
    qkv[..., : H * D] *= 14.0
    qkv = qkv.to(torch.bfloat16).cuda().contiguous()
    got = _attention_fused_rows(qkv)
    x = qkv.cpu().double().view(2, N, 3, H, D)
    want = F.scaled_dot_product_attention(x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2))
    want = want.transpose(1, 2).reshape(2, N, H * D)
    assert torch.isfinite(got.float()).all()
    assert (got.cpu().double() - want).abs().max().item() <= 4e-2          # near-one-hot rows: outputs up to ~4, bf16 steps of 2^-6


# ---- the short-key cross-attention kernel after its round-6 rework (unscaled Q, mask addends, blocks walking across the batch
# elements that share a K / V): /root/reference/diffsim/hacked_attn.py:74-81 with the 77-key prompt context ------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Bn,Bkv,Hn,Nq,Nk,Dn", [
    (12, 2, 8, 256, 77, 160),       # the 16 x 16 level: two blocks per batch element, a workgroup walks across batch elements
    (5, 2, 8, 200, 77, 160),        # batch not a multiple of the K / V count, ragged last query block
    (7, 3, 4, 300, 65, 40),         # the ragged third key block at its shortest ...
    (4, 2, 4, 130, 80, 80),         # ... and at its longest (the K80 instantiation's bounds)
    (3, 1, 2, 128, 64, 40), (3, 3, 2, 160, 81, 80), (6, 2, 5, 1024, 77, 64),
    (40, 2, 8, 1024, 77, 80)])      # more chunks than one round of workgroups
def test_short_key_attention_walks_batches(eng, dtype, Bn, Bkv, Hn, Nq, Nk, Dn):
    g = torch.Generator().manual_seed(Bn * 1000 + Nq + Nk + Dn)
    q = (torch.randn(Bn, Nq, Hn * Dn, generator=g) * 1.3).to(dtype)
    k = (torch.randn(Bkv, Nk, Hn * Dn, generator=g) * 1.3).to(dtype)
    v = torch.randn(Bkv, Nk, Hn * Dn, generator=g).to(dtype)
    idx = torch.arange(Bn) % Bkv
    want = F.scaled_dot_product_attention(q.double().view(Bn, Nq, Hn, Dn).transpose(1, 2),
                                          k.double()[idx].view(Bn, Nk, Hn, Dn).transpose(1, 2),
                                          v.double()[idx].view(Bn, Nk, Hn, Dn).transpose(1, 2)).transpose(1, 2).reshape(Bn, Nq, Hn * Dn)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    got = eng.op_attention(qd, kd, vd, Hn)
    assert torch.isfinite(got.float()).all()
    atol, rtol = (4e-3, 1.6e-2) if dtype == torch.bfloat16 else (5e-4, 2e-3)
    excess = ((got.cpu().double() - want).abs() / (atol + rtol * want.abs())).max().item()
    assert excess <= 1.0, excess
    assert torch.equal(got, eng.op_attention(qd, kd, vd, Hn))
    # a batch element's rows do not depend on the batch: element b alone, with its own K / V
    b = Bn - 1
    alone = eng.op_attention(qd[b:b + 1].contiguous(), kd[b % Bkv:b % Bkv + 1].contiguous(), vd[b % Bkv:b % Bkv + 1].contiguous(), Hn)
    assert torch.equal(got[b], alone[0])
