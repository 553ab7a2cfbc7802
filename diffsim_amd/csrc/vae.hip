// VAE encoder executor (AutoencoderKL.encode) and its C ABI -- SURVEY.md section 8f row 1.
//
// Replaces `pipe.vae.encode(image)` of the reference (diffsim/diffsim.py:92-96).  The arithmetic
// lives in un-vendored diffusers; it is restated from SURVEY.md Appendix A item 11:
//   conv_in 3->C0 | 4 x DownEncoderBlock2D (2 ResnetBlock2D without time embedding, GroupNorm eps 1e-6;
//   downsample = pad(0,1,0,1) + 3x3 stride-2 conv) | mid: resnet, 1-head attention with GroupNorm
//   and residual, resnet | GroupNorm + SiLU + conv_out -> quant_conv 1x1 -> (mean, logvar).
// Kernel reuse: every conv is the implicit-GEMM MFMA kernel (the stride-2 form with pad = 0),
// GroupNorm(+SiLU) is the two-pass HBM-bound kernel, conv_in is the direct small-K kernel.  The
// mid-block attention has one 512-wide head, too wide for the register-resident flash kernel, so
// it runs as three GEMMs per image around a row-softmax: S = q k^T, P = softmax(S/sqrt(C)),
// O = P v (+ b_v as a column bias: rows of P sum to one).  quant_conv is folded into conv_out at
// finalize (both are linear): W' = Wq Wco, b' = Wq bco + bq.
// Sampling z = mean + exp(0.5*clamp(logvar)) * eps stays on the host side (it consumes the
// caller's CPU generator in the reference's draw order).
#include <string>

#include "common.h"
#include "store.h"

using namespace dsim;

constexpr int VAE_VREP = 16;          // copies of the mid-block attention's to_v weight (finalize): images per batched v^T launch

struct dsim_vae : WeightStore {
    dsim_vae_cfg cfg;
};

namespace {

// W'[o][k] = sum_c Wq[o][c] * Wco[c][k]   (Wco packed [Cm][K] compute dtype; Wq f32 [Cm][Cm])
template <typename T>
__global__ void fold_quant_kernel(const float* __restrict__ wq, const T* __restrict__ wco, T* __restrict__ out, int Cm,
                                  int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y;
    if (k >= K) return;
    float acc = 0.f;
    for (int c = 0; c < Cm; ++c) acc = fmaf(wq[o * Cm + c], (float)wco[(size_t)c * K + k], acc);
    out[(size_t)o * K + k] = (T)acc;
}
__global__ void fold_quant_bias_kernel(const float* wq, const float* bco, const float* bq, float* out, int Cm) {
    const int o = threadIdx.x;
    if (o >= Cm) return;
    float acc = bq[o];
    for (int c = 0; c < Cm; ++c) acc = fmaf(wq[o * Cm + c], bco[c], acc);
    out[o] = acc;
}
// token-major [n][hw][C] compute dtype -> f32 NCHW [n][C][hw]
template <typename T>
__global__ void to_nchw_f32_kernel(const T* __restrict__ x, float* __restrict__ out, int HW, int Cc, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cc);
    const size_t t = i / Cc;
    const int p = (int)(t % HW);
    const size_t n = t / HW;
    out[(n * Cc + c) * HW + p] = (float)x[i];
}

struct VWalk {
    dsim_vae* h;
    Arena* ar;
    hipStream_t s;
    int n;                  // images
    bool run;
    void* gn_scratch = nullptr;

    size_t es() const { return dtype_size(h->dt); }
    void* alloc_act(size_t elems) { return ar->alloc(elems * es()); }

#define VGET(var, key)                                   \
    const Packed* var = h->find(key);                    \
    if (!var) return DSIM_ERR_MISSING_WEIGHT;

    // per-launch HIP-event brackets of a profiled encode (dsim_vae_profile_*), same record format as the U-Net executor's
    void pbegin(const std::string& name, double flops, double bytes) { if (run && h->profiling) prof_begin(h, s, name, flops, bytes); }
    void pend() { if (run && h->profiling) prof_end(h, s); }
    const char* dtn() const { return h->dt == DSIM_F32 ? "f32" : (h->dt == DSIM_F16 ? "f16" : "bf16"); }

    int gemm(GemmArgs& g) {
        g.zero_page = h->zero_page;
        if (!run) return DSIM_OK;
        if (h->profiling) {
            double fl, by;
            const std::string nm = gemm_family(g, h->dt, &fl, &by);
            pbegin(nm, fl, by);
        }
        const int st = launch_gemm(g, h->dt, s);
        pend();
        return st;
    }
    int linear(const void* a, int K, const void* w, const float* bias, const void* residual, void* out, int M, int N) {
        GemmArgs g;
        g.A0 = a; g.C0 = K; g.mode = GEMM_LINEAR; g.M = M; g.N = N; g.K = K; g.W = w; g.bias = bias;
        g.epi = residual ? EPI_RESIDUAL : EPI_NONE; g.residual = residual; g.out = out; g.ldo = N;
        return gemm(g);
    }
    // The 128 x 128 level (one image = 64 ... 128 of the 256-row tiles): its convs ask for those tiles at EVERY batch size
    // (GemmArgs.force_big), so that the statistics can come from the epilogue there as well -- a single image then runs that level's
    // convs on half the chip (+0.1 ms of a ~12 ms encode), every pair or chunk as before
    static int big_tiles(int hw, int Cout) { return hw >= 128 * 128 && hw % 256 == 0 && Cout % 256 == 0; }
    // GroupNorm statistics of a conv's output from its epilogue (GemmArgs.gn_part) where ONE image alone fills the chip's 256-row
    // tiles (the 512 x 512 and 256 x 256 levels: the same tiles at every batch size, so the numbers do not depend on the batch);
    // `stats` receives the partial buffer for the GroupNorm that consumes the output (null: that GroupNorm runs its own pass)
    int conv3(const Act& x, const Packed* w, const float* bias, const void* residual, void* out, int Cout, int stride,
              int pad, int batch, float** stats = nullptr) {
        GemmArgs g;
        g.A0 = x.p; g.C0 = x.C; g.mode = GEMM_CONV3; g.Hin = x.H; g.Win = x.W;
        g.Hout = stride == 2 ? x.H / 2 : x.H; g.Wout = stride == 2 ? x.W / 2 : x.W;
        g.stride = stride; g.ups = 0; g.pad = pad;
        g.M = batch * g.Hout * g.Wout; g.N = Cout; g.K = 9 * x.C; g.W = w->p; g.bias = bias;
        g.epi = residual ? EPI_RESIDUAL : EPI_NONE; g.residual = residual; g.out = out; g.ldo = Cout;
        if (stats) {
            *stats = nullptr;
            const int hw = g.Hout * g.Wout, cpg = Cout / h->cfg.norm_num_groups;
            g.force_big = big_tiles(hw, Cout);
            GemmArgs one = g;
            one.M = hw;                                          // the geometry of a single image
            if (h->cfg.norm_num_groups == 32 && Cout % 32 == 0 && (cpg == 4 || cpg == 8 || cpg == 16) && gemm_gn_stats_tile(one, h->dt)) {
                *stats = (float*)ar->alloc((size_t)batch * (hw / 64) * (Cout / 4) * 2 * sizeof(float));
                g.gn_part = *stats;
                g.gn_hw = hw;
            } else {
                g.force_big = 0;
            }
        }
        return gemm(g);
    }
    int gn(const Act& x, const Packed* g, const Packed* b, void* out, int silu, const float* stats = nullptr) {
        if (!run) return DSIM_OK;
        const int HW = x.H * x.W;
        if (stats) {
            pbegin(std::string("groupnorm_pre_") + dtn() + "|B" + std::to_string(n) + " HW" + std::to_string(HW) + " C" + std::to_string(x.C), 0.0,
                   2.0 * n * HW * (double)x.C * es());
            const int st = launch_groupnorm_pre(x.p, x.C, (const float*)g->p, (const float*)b->p, out, n, HW, h->cfg.norm_num_groups, 1e-6f,
                                                silu, h->dt, gn_scratch, stats, HW / 64, s);
            pend();
            return st;
        }
        pbegin(std::string("groupnorm_") + dtn() + "|B" + std::to_string(n) + " HW" + std::to_string(HW) + " C" + std::to_string(x.C), 0.0,
               (double)groupnorm_passes(x.C, 0, HW, h->cfg.norm_num_groups, h->dt) * n * HW * (double)x.C * es());
        const int st = launch_groupnorm(x.p, x.C, nullptr, 0, (const float*)g->p, (const float*)b->p, out, n, HW,
                                        h->cfg.norm_num_groups, 1e-6f, silu, h->dt, gn_scratch, s);
        pend();
        return st;
    }

    // in_stats: epilogue statistics of x (from the conv that produced it) for norm1; out_stats: receives those of this block's output
    int resnet(const std::string& p, const Act& x, int Cout, Act* out, const float* in_stats = nullptr, float** out_stats = nullptr) {
        const int Cin = x.C, M = n * x.H * x.W;
        VGET(n1w, p + "norm1.weight"); VGET(n1b, p + "norm1.bias");
        VGET(c1w, p + "conv1.weight"); VGET(c1b, p + "conv1.bias");
        VGET(n2w, p + "norm2.weight"); VGET(n2b, p + "norm2.bias");
        VGET(c2w, p + "conv2.weight"); VGET(c2b, p + "conv2.bias");
        out->p = alloc_act((size_t)M * Cout); out->C = Cout; out->H = x.H; out->W = x.W;
        // (the output's statistics buffer outlives this block's scratch: allocated before the mark)
        float* ostat = nullptr;
        {
            GemmArgs one;
            one.mode = GEMM_CONV3; one.Hout = x.H; one.Wout = x.W; one.Hin = x.H; one.Win = x.W; one.C0 = Cout; one.M = x.H * x.W; one.N = Cout;
            one.K = 9 * Cout; one.epi = EPI_RESIDUAL;
            one.force_big = big_tiles(x.H * x.W, Cout);
            const int cpg = Cout / h->cfg.norm_num_groups;
            if (out_stats && h->cfg.norm_num_groups == 32 && Cout % 32 == 0 && (cpg == 4 || cpg == 8 || cpg == 16) && gemm_gn_stats_tile(one, h->dt))
                ostat = (float*)ar->alloc((size_t)n * (x.H * x.W / 64) * (Cout / 4) * 2 * sizeof(float));
        }
        const size_t mk = ar->mark();
        Act t1{alloc_act((size_t)M * Cin), Cin, x.H, x.W};
        CK(gn(x, n1w, n1b, t1.p, 1, in_stats));
        Act t2{alloc_act((size_t)M * Cout), Cout, x.H, x.W};
        float* st2 = nullptr;
        CK(conv3(t1, c1w, (const float*)c1b->p, nullptr, t2.p, Cout, 1, 1, n, &st2));
        Act t3{t1.p, Cout, x.H, x.W};
        if (Cout > Cin) t3.p = alloc_act((size_t)M * Cout);
        CK(gn(t2, n2w, n2b, t3.p, 1, st2));
        const void* res = x.p;
        if (Cin != Cout) {
            VGET(scw, p + "conv_shortcut.weight"); VGET(scb, p + "conv_shortcut.bias");
            void* sc = t2.p;                                   // t2 is dead after norm2
            CK(linear(x.p, Cin, scw->p, (const float*)scb->p, nullptr, sc, M, Cout));
            res = sc;
        }
        {
            // conv2 + residual; its epilogue statistics (for the next block's norm1) go to the buffer reserved above
            GemmArgs g;
            g.A0 = t3.p; g.C0 = t3.C; g.mode = GEMM_CONV3; g.Hin = g.Hout = x.H; g.Win = g.Wout = x.W; g.stride = 1; g.ups = 0; g.pad = 1;
            g.M = M; g.N = Cout; g.K = 9 * t3.C; g.W = c2w->p; g.bias = (const float*)c2b->p;
            g.epi = EPI_RESIDUAL; g.residual = res; g.out = out->p; g.ldo = Cout;
            if (ostat) { g.gn_part = ostat; g.gn_hw = x.H * x.W; g.force_big = big_tiles(x.H * x.W, Cout); }
            CK(gemm(g));
        }
        if (out_stats) *out_stats = ostat;
        ar->release(mk);
        return DSIM_OK;
    }

    int attention(const std::string& p, const Act& x, Act* out) {
        const int C = x.C, N = x.H * x.W, M = n * N;
        VGET(gw, p + "group_norm.weight"); VGET(gb, p + "group_norm.bias");
        VGET(wq, p + "to_q.weight"); VGET(bq, p + "to_q.bias");
        VGET(wk, p + "to_k.weight"); VGET(bk, p + "to_k.bias");
        VGET(wv, p + "to_v.weight"); VGET(bv, p + "to_v.bias");
        VGET(wo, p + "to_out.0.weight"); VGET(bo, p + "to_out.0.bias");
        out->p = alloc_act((size_t)M * C); out->C = C; out->H = x.H; out->W = x.W;
        const size_t mk = ar->mark();
        char* t = (char*)alloc_act((size_t)M * C);
        CK(gn(x, gw, gb, t, 0));
        char* q = (char*)alloc_act((size_t)M * C);
        char* k = (char*)alloc_act((size_t)M * C);
        char* o = (char*)alloc_act((size_t)M * C);
        CK(linear(t, C, wq->p, (const float*)bq->p, nullptr, q, M, C));
        CK(linear(t, C, wk->p, (const float*)bk->p, nullptr, k, M, C));
        // Images per group: the score matrices of a whole group come from ONE batched launch each way (GemmArgs.wb_rows: row block i
        // multiplies image i's k / v^T), so the 32-tile P v of a single 4096-token image no longer runs as 16 x n quarter-chip
        // launches; same tiles' arithmetic bit for bit (gemm_skinny_kernel == gemm_kernel, asserted in the tests), so an image's
        // moments do not depend on its group.  Token counts that are not whole 256-row tiles keep one image per launch.
        int grp = 1;
        if (N % 256 == 0) {
            const size_t lim = 0x7fffffffull / ((size_t)N * N * es());
            grp = (int)std::min<size_t>((size_t)n, lim < 1 ? 1 : lim);
        }
        void* sc = alloc_act((size_t)grp * N * N);             // the group's score matrices
        char* vT = (char*)alloc_act((size_t)grp * C * N);
        const size_t img = (size_t)N * C * es();
        for (int i0 = 0; i0 < n; i0 += grp) {
            const int gi = std::min(grp, n - i0);
            const Packed* wrep = h->find(p + "to_v.weight_rep");
            for (int i = 0; i < gi;) {                                                                // v^T = Wv x^T
                const int nb = (wrep && C % 256 == 0) ? std::min(gi - i, VAE_VREP) : 1;
                GemmArgs gv;
                gv.A0 = nb > 1 ? wrep->p : wv->p; gv.C0 = C; gv.mode = GEMM_LINEAR; gv.M = nb * C; gv.N = N; gv.K = C;
                gv.W = t + (i0 + i) * img; gv.epi = EPI_NONE; gv.out = vT + i * img; gv.ldo = N;
                if (nb > 1) { gv.wb_rows = C; gv.wb_stride = (unsigned)img; }
                CK(gemm(gv));
                i += nb;
            }
            GemmArgs g;
            g.A0 = q + i0 * img; g.C0 = C; g.mode = GEMM_LINEAR; g.M = gi * N; g.N = N; g.K = C; g.W = k + i0 * img;
            g.epi = EPI_NONE; g.out = sc; g.ldo = N; g.wb_rows = N; g.wb_stride = (unsigned)img;
            CK(gemm(g));                                                                              // S = q k^T
            if (run) {
                pbegin(std::string("softmax_rows_") + dtn() + "|N" + std::to_string(N), 0.0, 2.0 * gi * N * (double)N * es());
                const int st = launch_softmax_rows(sc, sc, gi * N, N, 1.0f / sqrtf((float)C), h->dt, s);
                pend();
                CK(st);
            }
            GemmArgs o2;
            o2.A0 = sc; o2.C0 = N; o2.mode = GEMM_LINEAR; o2.M = gi * N; o2.N = C; o2.K = N; o2.W = vT; o2.bias = (const float*)bv->p;
            o2.epi = EPI_NONE; o2.out = o + i0 * img; o2.ldo = C; o2.wb_rows = N; o2.wb_stride = (unsigned)img;
            CK(gemm(o2));                                                                             // O = P v + b_v
        }
        CK(linear(o, C, wo->p, (const float*)bo->p, x.p, out->p, M, C));
        ar->release(mk);
        return DSIM_OK;
    }

    int go(const float* images, int S, float* moments) {
        const dsim_vae_cfg& c = h->cfg;
        const int nl = c.n_levels, ch0 = c.block_out_channels[0];
        gn_scratch = ar->alloc(groupnorm_scratch_bytes(n, c.norm_num_groups));
        VGET(ciw, "encoder.conv_in.weight"); VGET(cib, "encoder.conv_in.bias");
        Act x{alloc_act((size_t)n * S * S * ch0), ch0, S, S};
        float* xstat = nullptr;                 // epilogue statistics of x, when its producer made them
        const bool rows = conv_in_rows_applies(c.in_channels, S, ch0);
        if (rows && h->dt != DSIM_F32 && c.norm_num_groups == 32)       // (one 4-channel quad per group at 128 channels)
            xstat = (float*)ar->alloc((size_t)n * (S * S / 64) * (ch0 / 4) * 2 * sizeof(float));
        if (run) {
            pbegin(rows ? "conv_in_rows" : "prep_conv_in", 2.0 * n * S * S * (double)ch0 * 9 * c.in_channels,
                   (double)n * S * S * (ch0 * es() + c.in_channels * 4.0));
            const int st = rows ? conv_in_rows(images, (const float*)ciw->p, (const float*)cib->p, x.p, h->dt, n, S, xstat, s)
                                : prep_conv_in(images, nullptr, 1.f, 0.f, (const float*)ciw->p, (const float*)cib->p, x.p, h->dt, n,
                                               c.in_channels, S, ch0, 1, s);
            pend();
            CK(st);
        }
        for (int i = 0; i < nl; ++i) {
            const int co = c.block_out_channels[i];
            const std::string bp = "encoder.down_blocks." + std::to_string(i) + ".";
            for (int j = 0; j < c.layers_per_block; ++j) {
                Act r;
                float* rstat = nullptr;
                // (the last resnet of a level feeds the downsample conv, not a GroupNorm: no statistics asked of it)
                CK(resnet(bp + "resnets." + std::to_string(j) + ".", x, co, &r, xstat, j + 1 < c.layers_per_block ? &rstat : nullptr));
                x = r;
                xstat = rstat;
            }
            if (i != nl - 1) {
                VGET(dw, bp + "downsamplers.0.conv.weight"); VGET(db, bp + "downsamplers.0.conv.bias");
                Act d{alloc_act((size_t)n * (x.H / 2) * (x.W / 2) * co), co, x.H / 2, x.W / 2};
                CK(conv3(x, dw, (const float*)db->p, nullptr, d.p, co, 2, 0, n, &xstat));
                x = d;
            } else {
                xstat = nullptr;
            }
        }
        {
            const int cm = c.block_out_channels[nl - 1];
            Act r;
            CK(resnet("encoder.mid_block.resnets.0.", x, cm, &r)); x = r;
            CK(attention("encoder.mid_block.attentions.0.", x, &r)); x = r;
            CK(resnet("encoder.mid_block.resnets.1.", x, cm, &r)); x = r;
        }
        VGET(nw, "encoder.conv_norm_out.weight"); VGET(nb, "encoder.conv_norm_out.bias");
        VGET(cow, "encoder.conv_out_folded.weight"); VGET(cob, "encoder.conv_out_folded.bias");
        const int Cm = 2 * c.latent_channels, M = n * x.H * x.W;
        Act t{alloc_act((size_t)M * x.C), x.C, x.H, x.W};
        CK(gn(x, nw, nb, t.p, 1));
        void* mo = alloc_act((size_t)M * Cm);
        CK(conv3(t, cow, (const float*)cob->p, nullptr, mo, Cm, 1, 1, n));
        if (run) {
            const size_t total = (size_t)M * Cm;
            if (h->dt == DSIM_BF16)
                hipLaunchKernelGGL(to_nchw_f32_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                   (const bf16_t*)mo, moments, x.H * x.W, Cm, total);
            else if (h->dt == DSIM_F16)
                hipLaunchKernelGGL(to_nchw_f32_kernel<f16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                   (const f16_t*)mo, moments, x.H * x.W, Cm, total);
            else
                hipLaunchKernelGGL(to_nchw_f32_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                   (const float*)mo, moments, x.H * x.W, Cm, total);
            DSIM_HIP_CHECK(hipGetLastError());
        }
        return DSIM_OK;
    }
};

}  // namespace

extern "C" {

int dsim_vae_create(const dsim_vae_cfg* cfg, dsim_vae** out) {
    if (!cfg || !out) return DSIM_ERR_INVALID;
    if (cfg->n_levels < 1 || cfg->n_levels > DSIM_MAX_LEVELS) return DSIM_ERR_INVALID;
    if (cfg->compute_dtype != DSIM_F32 && cfg->compute_dtype != DSIM_BF16 && cfg->compute_dtype != DSIM_F16) return DSIM_ERR_INVALID;
    if (dsim_device_count() < 1) return DSIM_ERR_NO_DEVICE;
    dsim_vae* h = new dsim_vae();
    h->cfg = *cfg;
    h->dt = cfg->compute_dtype;
    if (h->dalloc(256, &h->zero_page) != DSIM_OK || hipMemset(h->zero_page, 0, 256) != hipSuccess) {
        dsim_vae_destroy(h);
        return DSIM_ERR_HIP;
    }
    *out = h;
    return DSIM_OK;
}

void dsim_vae_destroy(dsim_vae* h) {
    if (!h) return;
    h->free_all();
    delete h;
}

int dsim_vae_load_weight(dsim_vae* h, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim) {
    if (!h) return DSIM_ERR_INVALID;
    return h->add_raw(key, dev_ptr, dtype, shape, ndim);
}

int dsim_vae_finalize(dsim_vae* h, void* stream) {
    if (!h) return DSIM_ERR_INVALID;
    if (h->finalized) return DSIM_ERR_STATE;
    hipStream_t s = (hipStream_t)stream;
    CK(pack_all(h, s));
    // fold quant_conv (1x1, linear) into conv_out
    const Packed* cow = h->find("encoder.conv_out.weight");
    const Packed* cob = h->find("encoder.conv_out.bias");
    const Packed* qw = h->find("quant_conv.weight");
    const Packed* qb = h->find("quant_conv.bias");
    if (!cow || !cob || !qw || !qb) return DSIM_ERR_MISSING_WEIGHT;
    const int Cm = cow->rows, K = cow->cols;
    // quant_conv.weight was packed as a GEMM weight in the compute dtype; the fold wants it in f32
    float* wq32 = nullptr;
    CK(h->dalloc((size_t)Cm * Cm * 4, (void**)&wq32));
    auto it = h->raw.find("quant_conv.weight");
    if (it == h->raw.end()) return DSIM_ERR_MISSING_WEIGHT;
    CK(pack_vector(it->second.p, it->second.dtype, wq32, Cm * Cm, 0, s));
    Packed fw, fb;
    fw.rows = Cm; fw.cols = K;
    CK(h->dalloc((size_t)Cm * K * dtype_size(h->dt), &fw.p));
    CK(h->dalloc((size_t)Cm * 4, &fb.p));
    fb.rows = Cm; fb.cols = 1;
    if (h->dt == DSIM_BF16)
        hipLaunchKernelGGL(fold_quant_kernel<bf16_t>, dim3((K + 255) / 256, Cm), dim3(256), 0, s, wq32, (const bf16_t*)cow->p,
                           (bf16_t*)fw.p, Cm, K);
    else if (h->dt == DSIM_F16)
        hipLaunchKernelGGL(fold_quant_kernel<f16_t>, dim3((K + 255) / 256, Cm), dim3(256), 0, s, wq32, (const f16_t*)cow->p,
                           (f16_t*)fw.p, Cm, K);
    else
        hipLaunchKernelGGL(fold_quant_kernel<float>, dim3((K + 255) / 256, Cm), dim3(256), 0, s, wq32, (const float*)cow->p,
                           (float*)fw.p, Cm, K);
    hipLaunchKernelGGL(fold_quant_bias_kernel, dim3(1), dim3(64), 0, s, wq32, (const float*)cob->p, (const float*)qb->p,
                       (float*)fb.p, Cm);
    DSIM_HIP_CHECK(hipGetLastError());
    h->pk["encoder.conv_out_folded.weight"] = fw;
    h->pk["encoder.conv_out_folded.bias"] = fb;
    // the mid-block attention's to_v weight, VAE_VREP copies back to back: v^T = Wv x^T of a whole group of images is then ONE launch
    // whose row block i (= copy i of Wv) multiplies image i's tokens (GemmArgs.wb_rows)
    if (const Packed* wv = h->find("encoder.mid_block.attentions.0.to_v.weight")) {
        const size_t one = (size_t)wv->rows * wv->cols * dtype_size(h->dt);
        Packed rep;
        rep.rows = wv->rows * VAE_VREP; rep.cols = wv->cols;
        CK(h->dalloc(one * VAE_VREP, &rep.p));
        for (int i = 0; i < VAE_VREP; ++i)
            DSIM_HIP_CHECK(hipMemcpyAsync((char*)rep.p + i * one, wv->p, one, hipMemcpyDeviceToDevice, s));
        h->pk["encoder.mid_block.attentions.0.to_v.weight_rep"] = rep;
    }
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    h->raw.clear();
    h->finalized = true;
    Arena ar;
    VWalk w{h, &ar, s, 1, false};
    const int st = w.go(nullptr, 8 << (h->cfg.n_levels - 1), nullptr);
    if (st != DSIM_OK) { h->finalized = false; return st; }
    return DSIM_OK;
}

size_t dsim_vae_workspace_bytes(const dsim_vae* hc, int n_images, int image_size) {
    dsim_vae* h = const_cast<dsim_vae*>(hc);
    if (!h || !h->finalized || n_images < 1 || image_size < (1 << (h->cfg.n_levels - 1))) return 0;
    Arena ar;
    VWalk w{h, &ar, nullptr, n_images, false};
    if (w.go(nullptr, image_size, nullptr) != DSIM_OK) return 0;
    return ar.peak + 256;
}

int dsim_vae_encode(dsim_vae* h, const float* images, int n_images, int image_size, float* moments, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (!h || !images || !moments || !workspace || n_images < 1) return DSIM_ERR_INVALID;
    if (!h->finalized) return DSIM_ERR_STATE;
    if (image_size % (1 << (h->cfg.n_levels - 1))) return DSIM_ERR_INVALID;
    Arena ar;
    ar.dry = false;
    const uintptr_t b0 = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    const size_t lost = b0 - (uintptr_t)workspace;
    if (workspace_bytes < lost) return DSIM_ERR_WORKSPACE;
    ar.base = (char*)b0;
    ar.cap = workspace_bytes - lost;
    {
        Arena plan;
        VWalk pw{h, &plan, nullptr, n_images, false};
        CK(pw.go(nullptr, image_size, nullptr));
        if (plan.peak > ar.cap) return DSIM_ERR_WORKSPACE;
    }
    VWalk w{h, &ar, (hipStream_t)stream, n_images, true};
    CK(w.go(images, image_size, moments));
    return ar.overflow ? DSIM_ERR_WORKSPACE : DSIM_OK;
}

int dsim_vae_profile(dsim_vae* h, int enable) {
    if (!h) return DSIM_ERR_INVALID;
    h->clear_profile();
    h->profiling = enable != 0;
    return DSIM_OK;
}
int dsim_vae_profile_count(const dsim_vae* h) { return h ? (int)h->prof.size() : 0; }
int dsim_vae_profile_get(dsim_vae* h, int i, char* name, int name_cap, double* flops, double* bytes, double* ms) {
    return prof_get(h, i, name, name_cap, flops, bytes, ms);
}

int dsim_image_preprocess(const unsigned char* pixels_hwc, float* out, int n, int H, int W, int to_half, void* stream) {
    return image_preprocess(pixels_hwc, out, n, H, W, to_half != 0, (hipStream_t)stream);
}

int dsim_latent_sample(const float* moments, const float* eps, float* out, int n_out, int first, int stride, int C, int hw,
                       int eps_n, float scaling_factor, int round_fp16, void* stream) {
    return latent_sample(moments, eps, out, n_out, first, stride, C, hw, eps_n, scaling_factor, round_fp16 != 0, (hipStream_t)stream);
}

}  // extern "C"
