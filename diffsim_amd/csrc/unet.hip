// U-Net-to-tap graph executor and the C ABI of libdiffsim_amd.
//
// Implements, as a sequence of hand-written gfx950 kernels on ONE HIP stream, the sub-graph of
// diffusers' UNet2DConditionModel that the reference executes before its attention pre-hook
// fires (DiffSimPipeline.step -> self.unet(...), /root/reference/diffsim/diffsim_pipeline.py:213;
// hook at /root/reference/diffsim/diffsim.py:43-56).  Block control flow follows the reference's
// own restatement of it:
//   CrossAttnDownBlock2D  hacked_modules.py:537-620     UNetMidBlock2DCrossAttn  :622-688
//   CrossAttnUpBlock2D    hacked_modules.py:438-535     Transformer2DModel       :261-434
//   BasicTransformerBlock hacked_modules.py:17-136      q/k/v tap                hacked_attn.py:61-77
// and stops at the tap (nothing downstream of it feeds q/k/v).
//
// One `walk()` serves three purposes: PLAN (dry run: peak workspace bytes), RUN (launch), so the
// planner can never disagree with the executor.  Activations are token-major [B][HW][C] in the
// compute dtype; the workspace is a caller-provided arena with stack discipline.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "store.h"

using namespace dsim;

struct dsim_unet : WeightStore {
    dsim_unet_cfg cfg;
    int timestep = -1;
    float* temb = nullptr;          // [time_embed_dim]
    float* tscratch = nullptr;      // time-embedding scratch
    bool two_temb = false;          // SDXL: the CFG halves carry different time embeddings
    bool cfg_dedup = false;         // opt-in: compute the part of the graph that is identical in both CFG halves once
    int fusion = DSIM_FUSE_ALL;     // dsim_unet_set_fusion mask: which multi-op kernels replace their unfused chains
};

namespace {

// ---- the walk ----------------------------------------------------------------------------
struct Walk {
    dsim_unet* h;
    Arena* ar;
    hipStream_t s;
    int B2;                 // U-Net batch = 2 * images (CFG)
    bool run;               // false: plan only
    void* gn_scratch = nullptr;
    void* ctx_t = nullptr;  // [2][L][Dc] compute dtype
    void *q_out = nullptr, *k_out = nullptr, *v_out = nullptr;
    bool tapped = false;

    size_t es() const { return dtype_size(h->dt); }
    size_t max_tensor = 0;      // largest single activation (bytes): the kernels address tensors with 32-bit offsets
    void* alloc_act(size_t elems) {
        if (elems * es() > max_tensor) max_tensor = elems * es();
        return ar->alloc(elems * es());
    }

#define WGET(var, key)                                   \
    const Packed* var = h->find(key);                    \
    if (!var) return DSIM_ERR_MISSING_WEIGHT;

    // ---- optional per-launch HIP-event brackets (profiled forward only) --------------------
    void pbegin(const std::string& name, double flops, double bytes) {
        if (!run || !h->profiling) return;
        ProfRec r;
        r.name = name; r.flops = flops; r.bytes = bytes;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        (void)hipEventRecord(r.e0, s);
        h->prof.push_back(r);
    }
    void pend() {
        if (!run || !h->profiling) return;
        (void)hipEventRecord(h->prof.back().e1, s);
    }
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
    int linear(const void* a0, int c0, const void* a1, int c1, const Packed* w, const Packed* b, const void* residual,
               void* out, int M, int N, int ldo, int epi = -1) {
        GemmArgs g;
        g.A0 = a0; g.C0 = c0; g.A1 = a1; g.C1 = a1 ? c1 : 0;
        g.mode = GEMM_LINEAR;
        g.M = M; g.N = N; g.K = c0 + (a1 ? c1 : 0);
        g.W = w->p; g.bias = b ? (const float*)b->p : nullptr;
        g.epi = epi >= 0 ? epi : (residual ? EPI_RESIDUAL : EPI_NONE);
        if (g.epi == EPI_GEGLU) g.geglu_blk = geglu_block_rows(N);       // as pack_all() interleaved it
        g.residual = residual; g.out = out; g.ldo = ldo;
        return gemm(g);
    }
    int conv3(const Act& x, const Packed* w, const float* bias, const void* residual, void* out, int Cout, int stride,
              int ups, const float* bias_odd = nullptr) {
        GemmArgs g;
        if (h->two_temb && bias_odd) {
            g.bias2 = bias_odd;
            g.rows_per_batch = ups ? 4 * x.H * x.W : (stride == 2 ? ((x.H + 1) / 2) * ((x.W + 1) / 2) : x.H * x.W);
        }
        g.A0 = x.p; g.C0 = x.C; g.mode = GEMM_CONV3;
        g.Hin = x.H; g.Win = x.W;
        // stride 2, padding 1, kernel 3: (H - 1) / 2 + 1 rows, i.e. ceil(H / 2) (odd sides: --image_size 224 -> 28 -> 14 -> 7 -> 4)
        g.Hout = ups ? x.H * 2 : (stride == 2 ? (x.H + 1) / 2 : x.H);
        g.Wout = ups ? x.W * 2 : (stride == 2 ? (x.W + 1) / 2 : x.W);
        g.stride = stride; g.ups = ups;
        g.M = B2 * g.Hout * g.Wout; g.N = Cout; g.K = 9 * x.C;
        g.W = w->p; g.bias = bias;
        g.epi = residual ? EPI_RESIDUAL : EPI_NONE;
        g.residual = residual; g.out = out; g.ldo = Cout;
        return gemm(g);
    }
    int gn(const Act& x0, const Act* x1, const Packed* g, const Packed* b, void* out, float eps, int silu) {
        if (!run) return DSIM_OK;
        const double n = (double)B2 * x0.H * x0.W * (x0.C + (x1 ? x1->C : 0));
        pbegin(std::string("groupnorm_") + dtn() + "|B" + std::to_string(B2) + " HW" + std::to_string(x0.H * x0.W) + " C" +
                   std::to_string(x0.C + (x1 ? x1->C : 0)), 0.0,
               (double)groupnorm_passes(x0.C, x1 ? x1->C : 0, x0.H * x0.W, h->cfg.norm_num_groups, h->dt) * n * es());
        const int st = launch_groupnorm(x0.p, x0.C, x1 ? x1->p : nullptr, x1 ? x1->C : 0, (const float*)g->p,
                                        (const float*)b->p, out, B2, x0.H * x0.W, h->cfg.norm_num_groups, eps, silu,
                                        h->dt, gn_scratch, s);
        pend();
        return st;
    }
    int ln(const void* x, const Packed* g, const Packed* b, void* out, int M, int C) {
        if (!run) return DSIM_OK;
        pbegin(std::string("layernorm_") + dtn() + "|M" + std::to_string(M) + " C" + std::to_string(C), 0.0, 2.0 * M * (double)C * es());
        const int st = launch_layernorm(x, (const float*)g->p, (const float*)b->p, out, M, C, 1e-5f, h->dt, s);
        pend();
        return st;
    }
    int attn(const AttnArgs& a) {
        if (!run) return DSIM_OK;
        // (key sequences >= 2048 run the fixed-reference instantiation attn_kernel<T, D, true>: its own family)
        pbegin(std::string("attention_") + dtn() + "_d" + std::to_string(a.D) + attention_kernel_kind(a, h->dt) +
                   "|B" + std::to_string(a.B) + " H" + std::to_string(a.H) +
                   " Nq" + std::to_string(a.Nq) + " Nk" + std::to_string(a.Nk),
               4.0 * a.B * a.H * (double)a.Nq * a.Nk * a.D,
               (double)es() * a.B * a.H * a.D * (2.0 * a.Nq + 2.0 * a.Nk));
        const int st = launch_attention(a, h->dt, s);
        pend();
        return st;
    }

    // ResnetBlock2D; x1 = skip tensor concatenated after x0 on channels
    int resnet(const std::string& p, const Act& x0, const Act* x1, int Cout, Act* out) {
        const int Cin = x0.C + (x1 ? x1->C : 0), HW = x0.H * x0.W, M = B2 * HW;
        WGET(n1w, p + "norm1.weight"); WGET(n1b, p + "norm1.bias");
        WGET(c1w, p + "conv1.weight"); WGET(c1b, p + "conv1.bias_eff"); WGET(c1b2, p + "conv1.bias_eff2");
        WGET(n2w, p + "norm2.weight"); WGET(n2b, p + "norm2.bias");
        WGET(c2w, p + "conv2.weight"); WGET(c2b, p + "conv2.bias");
        out->p = alloc_act((size_t)M * Cout); out->C = Cout; out->H = x0.H; out->W = x0.W;
        const size_t mk = ar->mark();
        Act t1{alloc_act((size_t)M * Cin), Cin, x0.H, x0.W};
        CK(gn(x0, x1, n1w, n1b, t1.p, h->cfg.norm_eps, 1));
        Act t2{alloc_act((size_t)M * Cout), Cout, x0.H, x0.W};
        CK(conv3(t1, c1w, (const float*)c1b->p, nullptr, t2.p, Cout, 1, 0, (const float*)c1b2->p));
        Act t3{alloc_act((size_t)M * Cout), Cout, x0.H, x0.W};
        CK(gn(t2, nullptr, n2w, n2b, t3.p, h->cfg.norm_eps, 1));
        const void* res = x0.p;
        if (Cin != Cout) {
            WGET(scw, p + "conv_shortcut.weight"); WGET(scb, p + "conv_shortcut.bias");
            void* sc = alloc_act((size_t)M * Cout);
            CK(linear(x0.p, x0.C, x1 ? x1->p : nullptr, x1 ? x1->C : 0, scw, scb, nullptr, sc, M, Cout, Cout));
            res = sc;
        } else if (x1) {
            return DSIM_ERR_INVALID;
        }
        CK(conv3(t3, c2w, (const float*)c2b->p, res, out->p, Cout, 1, 0));
        ar->release(mk);
        return DSIM_OK;
    }

    int heads_at(int level) const { return h->cfg.heads_per_level[level] > 0 ? h->cfg.heads_per_level[level] : h->cfg.num_heads; }
    int depth_at(int level) const { return h->cfg.depth_per_level[level] > 0 ? h->cfg.depth_per_level[level] : 1; }

    // Transformer2DModel (GroupNorm -> proj_in -> `depth` BasicTransformerBlocks -> proj_out -> +residual; the
    // conv1x1 and the Linear form of proj_in/out are the same GEMM on token-major data).  tap_blk >= 0 stops
    // after norm1 of that transformer block and emits q,k,v (-2 = never, -1 = the last block).
    // half_in: x holds ONE copy per image (B2 / 2 batch elements, opt-in CFG de-duplication): everything up to the first
    // cross-attention -- the first place the prompt context enters -- runs on that half batch, then the residual stream, the
    // block input and the cross-attention query are duplicated into [image][cfg] order and the rest runs as usual.
    int transformer(const std::string& p, const Act& x, int level, int tap_blk, Act* out, bool half_in = false) {
        const int C = x.C, HW = x.H * x.W, M = B2 * HW, H = heads_at(level), D = C / H;
        const int Bfull = B2, Mh = (B2 / 2) * HW;
        const int L = h->cfg.ctx_len, Dc = h->cfg.cross_attention_dim;
        const int depth = depth_at(level);
        if (tap_blk == -1) tap_blk = depth - 1;
        if (tap_blk >= depth) return DSIM_ERR_INVALID;
        const bool tap_here = tap_blk >= 0;
        WGET(gnw, p + "norm.weight"); WGET(gnb, p + "norm.bias");
        WGET(piw, p + "proj_in.weight"); WGET(pib, p + "proj_in.bias");
        if (!tap_here) { out->p = alloc_act((size_t)M * C); out->C = C; out->H = x.H; out->W = x.W; }
        const size_t mk = ar->mark();
        void* t1 = alloc_act((size_t)M * C);
        void* hb = alloc_act((size_t)M * C);
        void* hbh = half_in ? alloc_act((size_t)Mh * C) : nullptr;      // half-batch residual stream / query / full-batch input copy
        void* abh = half_in ? alloc_act((size_t)Mh * C) : nullptr;
        void* xfull = half_in ? alloc_act((size_t)M * C) : nullptr;
        const void* xres = half_in ? xfull : x.p;                       // residual of proj_out: the block input, full batch
        if (half_in) {
            if (tap_blk == 0) return DSIM_ERR_INVALID;                  // (callers never de-duplicate a tapped first block)
            B2 = Bfull / 2;
        }
        const int M0 = half_in ? Mh : M;                                // rows of the part before the first cross-attention
        void* hb0 = half_in ? hbh : hb;
        CK(gn(x, nullptr, gnw, gnb, t1, 1e-6f, 0));
        CK(linear(t1, C, nullptr, 0, piw, pib, nullptr, hb0, M0, C, C));
        void* nb = t1;                                   // t1 is dead: reuse it for LayerNorm outputs
        void* big = nullptr;
        void* ab = nullptr;
        void* kvb = nullptr;
        for (int blk = 0; blk < depth; ++blk) {
        const std::string b = p + "transformer_blocks." + std::to_string(blk) + ".";
        WGET(l1w, b + "norm1.weight"); WGET(l1b, b + "norm1.bias");
        WGET(qkv, b + "attn1.qkv");
        const bool pre = half_in && blk == 0;            // still on the de-duplicated half batch
        const int Mx = pre ? Mh : M;
        void* hbx = pre ? hbh : hb;
        // LayerNorm + projection as one row-resident launch where the width has one (16-bit modes, C = 320; not the tapped block, whose
        // q, k, v go to three tensors)
        auto ln_proj = [&](const void* xin, const Packed* g, const Packed* be, const std::string& key, void* o, int Mr, int N) -> int {
            const auto it = h->pk.find(key);
            if (it == h->pk.end() || !(h->fusion & DSIM_FUSE_LNPROJ)) return 1;          // 1: run the unfused chain
            if (run) {
                RowLinArgs ra;
                ra.x = xin; ra.out = o; ra.ln_g = (const float*)g->p; ra.ln_b = (const float*)be->p; ra.stream = it->second.p;
                ra.M = Mr; ra.C = C; ra.N = N; ra.eps = 1e-5f; ra.dtype = h->dt;
                pbegin(std::string("ln_linear_") + dtn() + "|M" + std::to_string(Mr) + " N" + std::to_string(N) + " K" + std::to_string(C),
                       2.0 * Mr * (double)C * N, (double)Mr * (C + N) * es() + (double)C * N * es());
                const int st = launch_rowlin(ra, s);
                pend();
                if (st != DSIM_OK) return st;
            }
            return DSIM_OK;
        };
        const bool tapped_here = blk == tap_blk;
        int fq = 1;
        if (!tapped_here && !big) {
            // qkv [M][3C]; later the GEGLU output [M][4C] unless the feed-forward runs as one launch
            const bool ff1 = h->pk.count(b + "ff.stream") && (h->fusion & DSIM_FUSE_FF);
            big = alloc_act((size_t)M * (ff1 ? 3 : 4) * C);
            ab = alloc_act((size_t)M * C);
            kvb = alloc_act((size_t)2 * L * 2 * C);
        }
        if (!tapped_here) fq = ln_proj(hbx, l1w, l1b, b + "attn1.qkv.stream", big, Mx, 3 * C);
        if (fq < 0) return fq;
        if (fq > 0) CK(ln(hbx, l1w, l1b, nb, Mx, C));
        if (blk == tap_blk) {
            // hacked_attn.py:61-69: to_q / to_k / to_v, no bias; written [B][N][H*D]
            Packed wq = *qkv, wk = *qkv, wv = *qkv;
            wk.p = (char*)qkv->p + (size_t)C * C * es();
            wv.p = (char*)qkv->p + (size_t)2 * C * C * es();
            // one launch when q, k, v lie at equal distances (engine.py allocates them as one [3][...] buffer) and the width
            // tiles by 320: the packed [3C][C] weight as one N = 3C GEMM whose column runs go to the three tensors
            const long long qk = (const char*)k_out - (const char*)q_out, kv = (const char*)v_out - (const char*)k_out;
            const bool bm_split = C % 320 == 0;
            if ((h->fusion & DSIM_FUSE_TAPQKV) && qk == kv && qk >= (long long)M * C * (long long)es() && 3 * qk < 0x7fffffffll && bm_split) {
                GemmArgs g;
                g.A0 = nb; g.C0 = C; g.mode = GEMM_LINEAR; g.M = M; g.N = 3 * C; g.K = C;
                g.W = qkv->p; g.epi = EPI_NONE; g.out = q_out; g.ldo = C; g.out_split = C; g.out_split_stride = qk;
                CK(gemm(g));
            } else {
                CK(linear(nb, C, nullptr, 0, &wq, nullptr, nullptr, q_out, M, C, C));
                CK(linear(nb, C, nullptr, 0, &wk, nullptr, nullptr, k_out, M, C, C));
                CK(linear(nb, C, nullptr, 0, &wv, nullptr, nullptr, v_out, M, C, C));
            }
            tapped = true;
            ar->release(mk);
            return DSIM_OK;
        }
        WGET(o1w, b + "attn1.to_out.0.weight"); WGET(o1b, b + "attn1.to_out.0.bias");
        WGET(l2w, b + "norm2.weight"); WGET(l2b, b + "norm2.bias");
        WGET(q2w, b + "attn2.to_q.weight"); WGET(kv2, b + "attn2.kv");
        WGET(o2w, b + "attn2.to_out.0.weight"); WGET(o2b, b + "attn2.to_out.0.bias");
        WGET(l3w, b + "norm3.weight"); WGET(l3b, b + "norm3.bias");
        WGET(f1w, b + "ff.net.0.proj.weight"); WGET(f1b, b + "ff.net.0.proj.bias");
        WGET(f2w, b + "ff.net.2.weight"); WGET(f2b, b + "ff.net.2.bias");
        // self-attention
        if (fq > 0) CK(linear(nb, C, nullptr, 0, qkv, nullptr, nullptr, big, Mx, 3 * C, 3 * C));
        {
            AttnArgs a;
            a.q = big; a.ldq = 3 * C;
            a.k = (char*)big + (size_t)C * es(); a.v = (char*)big + (size_t)2 * C * es(); a.ldk = 3 * C;
            a.out = pre ? abh : ab; a.ldo = C; a.B = B2; a.Bkv = B2; a.H = H; a.Nq = HW; a.Nk = HW; a.D = D;
            CK(attn(a));
        }
        CK(linear(pre ? abh : ab, C, nullptr, 0, o1w, o1b, hbx, hbx, Mx, C, C));
        // cross-attention against the prompt context: batch element b uses ctx[b % 2]
        {
            const int f2 = ln_proj(hbx, l2w, l2b, b + "attn2.to_q.stream", pre ? abh : ab, Mx, C);
            if (f2 < 0) return f2;
            if (f2 > 0) {
                CK(ln(hbx, l2w, l2b, nb, Mx, C));
                CK(linear(nb, C, nullptr, 0, q2w, nullptr, nullptr, pre ? abh : ab, Mx, C, C));
            }
        }
        if (pre) {
            // the two CFG halves part here: [image] -> [image][cfg] for the residual stream, the query and the block input
            B2 = Bfull;
            if (run) {
                const size_t per = (size_t)HW * C * es();
                pbegin(std::string("cfg_duplicate_") + dtn(), 0.0, 3.0 * 3.0 * Mh * (double)C * es());
                int st = dup_batch(hbh, hb, Bfull / 2, per, s);
                if (st == DSIM_OK) st = dup_batch(abh, ab, Bfull / 2, per, s);
                if (st == DSIM_OK) st = dup_batch(x.p, xfull, Bfull / 2, per, s);
                pend();
                CK(st);
            }
        }
        CK(linear(ctx_t, Dc, nullptr, 0, kv2, nullptr, nullptr, kvb, 2 * L, 2 * C, 2 * C));
        {
            AttnArgs a;
            a.q = ab; a.ldq = C;
            a.k = kvb; a.v = (char*)kvb + (size_t)C * es(); a.ldk = 2 * C;
            a.out = big; a.ldo = C; a.B = B2; a.Bkv = 2; a.H = H; a.Nq = HW; a.Nk = L; a.D = D;
            CK(attn(a));
        }
        CK(linear(big, C, nullptr, 0, o2w, o2b, hb, hb, M, C, C));
        // feed-forward: norm3 -> Linear(C,8C) -> h*gelu(g) -> Linear(4C,C) -> + residual.  One row-resident launch where
        // the width has one (16-bit modes, C = 320); else LayerNorm, the GEGLU GEMM (h*gelu(g) in its epilogue) and ff.net.2.
        const auto fst = h->pk.find(b + "ff.stream");
        if (fst != h->pk.end() && (h->fusion & DSIM_FUSE_FF)) {
            if (run) {
                FFArgs fa;
                fa.x = hb; fa.out = hb; fa.ln_g = (const float*)l3w->p; fa.ln_b = (const float*)l3b->p;
                fa.stream = fst->second.p; fa.b1 = (const float*)f1b->p; fa.b2 = (const float*)f2b->p; fa.M = M; fa.C = C;
                fa.eps = 1e-5f; fa.dtype = h->dt;
                pbegin(std::string("ff_fused_") + dtn() + "|M" + std::to_string(M) + " C" + std::to_string(C),
                       2.0 * M * (double)C * 12 * C, 2.0 * M * (double)C * es() + 12.0 * C * C * es());
                const int st = launch_ff_fused(fa, s);
                pend();
                CK(st);
            }
        } else {
            CK(ln(hb, l3w, l3b, nb, M, C));
            CK(linear(nb, C, nullptr, 0, f1w, f1b, nullptr, big, M, 8 * C, 4 * C, EPI_GEGLU));
            CK(linear(big, 4 * C, nullptr, 0, f2w, f2b, hb, hb, M, C, C));
        }
        }   // transformer blocks
        WGET(pow_, p + "proj_out.weight"); WGET(pob, p + "proj_out.bias");
        CK(linear(hb, C, nullptr, 0, pow_, pob, xres, out->p, M, C, C));
        ar->release(mk);
        return DSIM_OK;
    }

    int go(const float* lat, const float* noise, float sa, float sb, const float* ctx) {
        const dsim_unet_cfg& c = h->cfg;
        const int nl = c.n_levels, S = c.sample_size, ch0 = c.block_out_channels[0];
        const int L = c.ctx_len, Dc = c.cross_attention_dim;
        gn_scratch = ar->alloc(groupnorm_scratch_bytes(B2, c.norm_num_groups));
        ctx_t = ar->alloc((size_t)2 * L * Dc * es());
        if (run) CK(convert_f32_to(ctx, ctx_t, h->dt, (size_t)2 * L * Dc, s));
        WGET(ciw, "conv_in.weight"); WGET(cib, "conv_in.bias");
        Act x{alloc_act((size_t)B2 * S * S * ch0), ch0, S, S};
        if (run) {
            pbegin("prep_conv_in", 2.0 * B2 * S * S * (double)ch0 * 9 * c.in_channels, (double)B2 * S * S * ch0 * es());
            const int st = prep_conv_in(lat, noise, sa, sb, (const float*)ciw->p, (const float*)cib->p, x.p, h->dt,
                                        B2 / 2, c.in_channels, S, ch0, 2, s);
            pend();
            CK(st);
        }
        std::vector<Act> skips;
        skips.push_back(x);
        // Opt-in CFG de-duplication (dsim_unet_set_cfg_dedup): the reference feeds torch.cat([latents] * 2) with [negative,
        // positive] prompt embeddings (diffsim_pipeline.py:208-221), so conv_in, the first ResnetBlock2D and the first
        // transformer up to its cross-attention query see two bit-identical batch halves.  With one time embedding for both
        // halves (SD1.5; SDXL's text_time embedding differs per half) and the tap outside that block, they are computed once
        // per image and duplicated where the prompt context first enters.  Same kernels, batch-invariant: same bits.
        const bool dedup = h->cfg_dedup && !h->two_temb && c.down_has_attn[0] && c.layers_per_block >= 1 &&
                           !(c.tap_block == DSIM_TAP_DOWN && c.tap_layer == 0);
        Act xh{nullptr, ch0, S, S};
        if (dedup) {
            xh.p = alloc_act((size_t)(B2 / 2) * S * S * ch0);
            if (run)
                CK(prep_conv_in(lat, noise, sa, sb, (const float*)ciw->p, (const float*)cib->p, xh.p, h->dt, B2 / 2, c.in_channels,
                                S, ch0, 1, s));
        }
        // ---- down path (hacked_modules.py:583-618) ---------------------------------------
        for (int i = 0; i < nl; ++i) {
            const int co = c.block_out_channels[i];
            const std::string bp = "down_blocks." + std::to_string(i) + ".";
            for (int j = 0; j < c.layers_per_block; ++j) {
                const bool half = dedup && i == 0 && j == 0;
                Act r;
                if (half) {
                    const int Bfull = B2;
                    B2 = Bfull / 2;
                    const int st = resnet(bp + "resnets.0.", xh, nullptr, co, &r);
                    B2 = Bfull;
                    CK(st);
                } else {
                    CK(resnet(bp + "resnets." + std::to_string(j) + ".", x, nullptr, co, &r));
                }
                x = r;
                if (c.down_has_attn[i]) {
                    const int ta = c.tap_attn < 0 ? c.layers_per_block - 1 : c.tap_attn;
                    const bool tap = c.tap_block == DSIM_TAP_DOWN && c.tap_layer == i && j == ta;
                    Act t;
                    CK(transformer(bp + "attentions." + std::to_string(j) + ".", x, i, tap ? c.tap_tfm : -2, &t, half));
                    if (tap) return DSIM_OK;
                    x = t;
                }
                skips.push_back(x);
            }
            if (i != nl - 1) {
                WGET(dw, bp + "downsamplers.0.conv.weight"); WGET(db, bp + "downsamplers.0.conv.bias");
                Act d{alloc_act((size_t)B2 * ((x.H + 1) / 2) * ((x.W + 1) / 2) * co), co, (x.H + 1) / 2, (x.W + 1) / 2};
                CK(conv3(x, dw, (const float*)db->p, nullptr, d.p, co, 2, 0));
                x = d;
                skips.push_back(x);
            }
        }
        // ---- mid (hacked_modules.py:632,664-675) -------------------------------------------
        {
            const int cm = c.block_out_channels[nl - 1];
            Act r;
            CK(resnet("mid_block.resnets.0.", x, nullptr, cm, &r));
            x = r;
            const bool tap = c.tap_block == DSIM_TAP_MID;
            Act t;
            CK(transformer("mid_block.attentions.0.", x, nl - 1, tap ? c.tap_tfm : -2, &t));
            if (tap) return DSIM_OK;
            x = t;
            CK(resnet("mid_block.resnets.1.", x, nullptr, cm, &r));
            x = r;
        }
        // ---- up path (hacked_modules.py:457-527) -------------------------------------------
        for (int i = 0; i < nl; ++i) {
            const int co = c.block_out_channels[nl - 1 - i];
            const std::string bp = "up_blocks." + std::to_string(i) + ".";
            const int nres = c.layers_per_block + 1;
            for (int j = 0; j < nres; ++j) {
                if (skips.empty()) return DSIM_ERR_INVALID;
                Act sk = skips.back();
                skips.pop_back();
                Act r;
                CK(resnet(bp + "resnets." + std::to_string(j) + ".", x, &sk, co, &r));
                x = r;
                if (c.up_has_attn[i]) {
                    const int ta = c.tap_attn < 0 ? nres - 1 : c.tap_attn;
                    const bool tap = c.tap_block == DSIM_TAP_UP && c.tap_layer == i && j == ta;
                    Act t;
                    CK(transformer(bp + "attentions." + std::to_string(j) + ".", x, nl - 1 - i, tap ? c.tap_tfm : -2, &t));
                    if (tap) return DSIM_OK;
                    x = t;
                }
            }
            if (i != nl - 1) {
                WGET(uw, bp + "upsamplers.0.conv.weight"); WGET(ub, bp + "upsamplers.0.conv.bias");
                // the upsampled size is the next skip's (diffusers' upsample_size, hacked_modules.py:531-533): twice the side
                // except below an odd level, where the nearest-neighbour resize to the explicit size runs as its own launch
                const int th = skips.empty() ? x.H * 2 : skips.back().H, tw = skips.empty() ? x.W * 2 : skips.back().W;
                Act u{alloc_act((size_t)B2 * th * tw * co), co, th, tw};
                if (th == 2 * x.H && tw == 2 * x.W) {
                    CK(conv3(x, uw, (const float*)ub->p, nullptr, u.p, co, 1, 1));      // x2 folded into the conv's gather
                } else {
                    Act rz{alloc_act((size_t)B2 * th * tw * co), co, th, tw};
                    if (run) {
                        pbegin(std::string("resize_nearest_") + dtn(), 0.0, 2.0 * B2 * th * tw * (double)co * es());
                        const int st = resize_nearest(x.p, rz.p, B2, x.H, x.W, th, tw, (size_t)co * es(), s);
                        pend();
                        CK(st);
                    }
                    CK(conv3(rz, uw, (const float*)ub->p, nullptr, u.p, co, 1, 0));
                }
                x = u;
            }
        }
        return DSIM_ERR_INVALID;   // the tap was never reached: bad tap_block / tap_layer
    }
};

int tap_geometry(const dsim_unet_cfg& c, int* tokens, int* heads, int* hd) {
    const int nl = c.n_levels;
    int level;   // resolution level of the tapped block
    if (c.tap_block == DSIM_TAP_DOWN) {
        if (c.tap_layer < 0 || c.tap_layer >= nl || !c.down_has_attn[c.tap_layer]) return DSIM_ERR_INVALID;
        if (c.tap_attn >= c.layers_per_block) return DSIM_ERR_INVALID;
        level = c.tap_layer;
    } else if (c.tap_block == DSIM_TAP_MID) {
        if (c.tap_attn > 0) return DSIM_ERR_INVALID;
        level = nl - 1;
    } else if (c.tap_block == DSIM_TAP_UP) {
        const int i = c.tap_layer;
        if (i < 0 || i >= nl || !c.up_has_attn[i]) return DSIM_ERR_INVALID;
        if (c.tap_attn > c.layers_per_block) return DSIM_ERR_INVALID;
        level = nl - 1 - i;
    } else {
        return DSIM_ERR_INVALID;
    }
    const int depth = c.depth_per_level[level] > 0 ? c.depth_per_level[level] : 1;
    if (c.tap_tfm >= depth) return DSIM_ERR_INVALID;
    // spatial side at that level: one downsample per level except after the last
    int side = c.sample_size;
    for (int l = 0; l < level; ++l) side = (side + 1) / 2;        // stride-2 convs: ceil (odd sides)
    const int C = c.block_out_channels[level];
    const int H = c.heads_per_level[level] > 0 ? c.heads_per_level[level] : c.num_heads;
    *tokens = side * side;
    *heads = H;
    *hd = C / H;
    return DSIM_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int dsim_version(void) { return DSIM_ABI_VERSION; }

const char* dsim_strerror(int st) {
    switch (st) {
        case DSIM_OK: return "ok";
        case DSIM_ERR_INVALID: return "invalid argument or unsupported shape";
        case DSIM_ERR_MISSING_WEIGHT: return "a parameter needed before the tap was never loaded";
        case DSIM_ERR_WORKSPACE: return "workspace too small";
        case DSIM_ERR_HIP: return "HIP runtime error";
        case DSIM_ERR_STATE: return "call order violated";
        case DSIM_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

int dsim_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dsim_unet_create(const dsim_unet_cfg* cfg, dsim_unet** out) {
    if (!cfg || !out) return DSIM_ERR_INVALID;
    if (cfg->n_levels < 1 || cfg->n_levels > DSIM_MAX_LEVELS) return DSIM_ERR_INVALID;
    if (cfg->compute_dtype != DSIM_F32 && cfg->compute_dtype != DSIM_BF16 && cfg->compute_dtype != DSIM_F16) return DSIM_ERR_INVALID;
    int t, hh, d;
    CK(tap_geometry(*cfg, &t, &hh, &d));
    if (dsim_device_count() < 1) return DSIM_ERR_NO_DEVICE;
    dsim_unet* h = new dsim_unet();
    h->cfg = *cfg;
    h->dt = cfg->compute_dtype;
    if (h->dalloc(256, &h->zero_page) != DSIM_OK || hipMemset(h->zero_page, 0, 256) != hipSuccess) {
        dsim_unet_destroy(h);
        return DSIM_ERR_HIP;
    }
    *out = h;
    return DSIM_OK;
}

void dsim_unet_destroy(dsim_unet* h) {
    if (!h) return;
    for (void* p : h->owned) (void)hipFree(p);
    delete h;
}

int dsim_unet_load_weight(dsim_unet* h, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim) {
    if (!h || !key || !dev_ptr || !shape || ndim < 1 || ndim > 4) return DSIM_ERR_INVALID;
    if (dtype != DSIM_F32 && dtype != DSIM_BF16 && dtype != DSIM_F16) return DSIM_ERR_INVALID;
    if (h->finalized) return DSIM_ERR_STATE;
    RawW w;
    w.p = dev_ptr; w.dtype = dtype; w.shape.assign(shape, shape + ndim);
    h->raw[key] = w;
    return DSIM_OK;
}

int dsim_unet_finalize(dsim_unet* h, void* stream) {
    if (!h) return DSIM_ERR_INVALID;
    if (h->finalized) return DSIM_ERR_STATE;
    hipStream_t s = (hipStream_t)stream;
    CK(pack_all(h, s));
    const int ted = h->cfg.block_out_channels[0] * 4;
    const int addin = h->cfg.addition_embed ? h->cfg.pooled_dim + 6 * h->cfg.addition_time_embed_dim : 0;
    CK(h->dalloc((size_t)2 * ted * 4, (void**)&h->temb));
    CK(h->dalloc((size_t)(h->cfg.block_out_channels[0] + 5 * ted + addin + 64) * 4, (void**)&h->tscratch));
    // conv1.bias_eff / bias_eff2 (conv1.bias + time_emb_proj(silu(temb)) for the even / odd CFG half) are
    // filled by set_timestep / set_conditioning
    std::vector<std::string> res;
    for (auto& kv : h->pk)
        if (ends_with(kv.first, "time_emb_proj.weight")) res.push_back(kv.first);
    for (auto& k : res) {
        const std::string p = k.substr(0, k.size() - strlen("time_emb_proj.weight"));
        for (const char* suffix : {"conv1.bias_eff", "conv1.bias_eff2"}) {
            Packed P;
            P.rows = h->pk[k].rows; P.cols = 1;
            CK(h->dalloc((size_t)P.rows * 4, &P.p));
            h->pk[p + suffix] = P;
        }
    }
    // fused feed-forward (rowres.hip): one weight stream per transformer block of a width the kernel covers (16-bit modes only)
    if (h->dt != DSIM_F32) {
        std::vector<std::string> ffs;
        for (auto& kv : h->pk)
            if (ends_with(kv.first, "ff.net.0.proj.weight") && ff_stream_bytes(kv.second.cols)) ffs.push_back(kv.first);
        for (auto& k : ffs) {
            const std::string b = k.substr(0, k.size() - strlen("ff.net.0.proj.weight"));
            const Packed* w2 = h->find(b + "ff.net.2.weight");
            if (!w2) return DSIM_ERR_MISSING_WEIGHT;
            const int C = h->pk[k].cols;
            Packed P;
            P.rows = 1; P.cols = C; P.bytes = ff_stream_bytes(C);
            CK(h->dalloc(P.bytes, &P.p));
            CK(pack_ff_stream(h->pk[k].p, w2->p, P.p, C, s));
            h->pk[b + "ff.stream"] = P;
        }
        // LayerNorm-fronted projections of the same blocks: norm1 -> to_q|to_k|to_v and norm2 -> attn2.to_q
        std::vector<std::pair<std::string, std::string>> lps;
        for (auto& kv : h->pk) {
            if (ends_with(kv.first, "attn1.qkv") && rowlin_stream_bytes(kv.second.cols, kv.second.rows))
                lps.push_back({kv.first, kv.first + ".stream"});
            if (ends_with(kv.first, "attn2.to_q.weight") && rowlin_stream_bytes(kv.second.cols, kv.second.rows))
                lps.push_back({kv.first, kv.first.substr(0, kv.first.size() - strlen("weight")) + "stream"});
        }
        for (auto& kn : lps) {
            const Packed& W = h->pk[kn.first];
            Packed P;
            P.rows = W.rows; P.cols = W.cols; P.bytes = rowlin_stream_bytes(W.cols, W.rows);
            CK(h->dalloc(P.bytes, &P.p));
            CK(pack_rowlin_stream(W.p, P.p, W.cols, W.rows, s));
            h->pk[kn.second] = P;
        }
    }
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    h->raw.clear();
    h->finalized = true;
    // make sure every parameter up to the tap exists: dry walk
    Arena ar;
    Walk w{h, &ar, s, 2, false};
    const int st = w.go(nullptr, nullptr, 0.f, 0.f, nullptr);
    if (st != DSIM_OK) { h->finalized = false; return st; }
    return DSIM_OK;
}

static int set_cond(dsim_unet* h, int t, const float* text_embeds, const float* time_ids, hipStream_t s) {
    const dsim_unet_cfg& c = h->cfg;
    const int ch0 = c.block_out_channels[0], ted = ch0 * 4;
    const bool add = c.addition_embed != 0;
    if (add && (!text_embeds || !time_ids)) return DSIM_ERR_INVALID;
    float* emb = h->tscratch;              // [ch0]
    float* h1 = emb + ch0;                 // [ted]
    float* tp = h1 + ted;                  // [ted] per-resnet projection
    float* base = tp + ted;                // [ted] time_embedding(t)
    float* a1 = base + ted;                // [ted]
    float* aug = a1 + ted;                 // [ted]
    float* ain = aug + ted;                // [pooled + 6*atd]
    const Packed* w1 = h->find("time_embedding.linear_1.weight");
    const Packed* b1 = h->find("time_embedding.linear_1.bias");
    const Packed* w2 = h->find("time_embedding.linear_2.weight");
    const Packed* b2 = h->find("time_embedding.linear_2.bias");
    if (!w1 || !b1 || !w2 || !b2) return DSIM_ERR_MISSING_WEIGHT;
    CK(timestep_sincos(emb, ch0, t, s));
    CK(gemv_f32(w1->p, DSIM_F32, b1->p, DSIM_F32, emb, h1, ted, ch0, 0, s));
    CK(gemv_f32(w2->p, DSIM_F32, b2->p, DSIM_F32, h1, base, ted, ted, 1, s));
    for (int half = 0; half < 2; ++half) {
        float* temb = h->temb + (size_t)half * ted;
        if (add) {
            // UNet2DConditionModel "text_time": cat(text_embeds, Timesteps(time_ids.flatten())) -> add_embedding
            const Packed* aw1 = h->find("add_embedding.linear_1.weight");
            const Packed* ab1 = h->find("add_embedding.linear_1.bias");
            const Packed* aw2 = h->find("add_embedding.linear_2.weight");
            const Packed* ab2 = h->find("add_embedding.linear_2.bias");
            if (!aw1 || !ab1 || !aw2 || !ab2) return DSIM_ERR_MISSING_WEIGHT;
            const int P = c.pooled_dim, atd = c.addition_time_embed_dim, nin = P + 6 * atd;
            DSIM_HIP_CHECK(hipMemcpyAsync(ain, text_embeds + (size_t)half * P, (size_t)P * 4, hipMemcpyDeviceToDevice, s));
            CK(sincos_values(ain + P, atd, time_ids + (size_t)half * 6, 6, s));
            CK(gemv_f32(aw1->p, DSIM_F32, ab1->p, DSIM_F32, ain, a1, ted, nin, 0, s));
            CK(gemv_f32(aw2->p, DSIM_F32, ab2->p, DSIM_F32, a1, aug, ted, ted, 1, s));
            CK(add_vectors_f32(base, aug, temb, ted, s));
        } else {
            DSIM_HIP_CHECK(hipMemcpyAsync(temb, base, (size_t)ted * 4, hipMemcpyDeviceToDevice, s));
        }
        for (auto& kv : h->pk) {
            if (!ends_with(kv.first, "time_emb_proj.weight")) continue;
            const std::string p = kv.first.substr(0, kv.first.size() - strlen("time_emb_proj.weight"));
            const Packed* tb = h->find(p + "time_emb_proj.bias");
            const Packed* cb = h->find(p + "conv1.bias");
            const Packed* eff = h->find(p + (half ? "conv1.bias_eff2" : "conv1.bias_eff"));
            if (!tb || !cb || !eff) return DSIM_ERR_MISSING_WEIGHT;
            const int n = kv.second.rows;
            if (n > ted) return DSIM_ERR_INVALID;
            CK(gemv_f32(kv.second.p, DSIM_F32, tb->p, DSIM_F32, temb, tp, n, ted, 1, s));
            CK(add_vectors_f32((const float*)cb->p, tp, (float*)eff->p, n, s));
        }
    }
    h->two_temb = add;
    h->timestep = t;
    return DSIM_OK;
}

int dsim_unet_set_timestep(dsim_unet* h, int t, void* stream) {
    if (!h || t < 0) return DSIM_ERR_INVALID;
    if (!h->finalized) return DSIM_ERR_STATE;
    if (h->cfg.addition_embed) return DSIM_ERR_STATE;       // SDXL graphs need set_conditioning
    return set_cond(h, t, nullptr, nullptr, (hipStream_t)stream);
}

int dsim_unet_set_conditioning(dsim_unet* h, int t, const float* text_embeds, const float* time_ids, void* stream) {
    if (!h || t < 0) return DSIM_ERR_INVALID;
    if (!h->finalized) return DSIM_ERR_STATE;
    return set_cond(h, t, text_embeds, time_ids, (hipStream_t)stream);
}

size_t dsim_unet_workspace_bytes(const dsim_unet* hc, int n_images) {
    dsim_unet* h = const_cast<dsim_unet*>(hc);
    if (!h || !h->finalized || n_images < 1) return 0;
    Arena ar;
    Walk w{h, &ar, nullptr, 2 * n_images, false};
    if (w.go(nullptr, nullptr, 0.f, 0.f, nullptr) != DSIM_OK) return 0;
    if (w.max_tensor >= 0x7fffffffull) return 0;      // a >= 2 GiB activation: the batch does not fit one call
    return ar.peak + 256;
}

int dsim_unet_tap_shape(const dsim_unet* h, int* tokens, int* heads, int* head_dim) {
    if (!h || !tokens || !heads || !head_dim) return DSIM_ERR_INVALID;
    return tap_geometry(h->cfg, tokens, heads, head_dim);
}

int dsim_unet_set_tap(dsim_unet* h, int tap_block, int tap_layer, int tap_attn, int tap_tfm) {
    if (!h) return DSIM_ERR_INVALID;
    if (!h->finalized) return DSIM_ERR_STATE;
    const dsim_unet_cfg old = h->cfg;
    h->cfg.tap_block = tap_block; h->cfg.tap_layer = tap_layer; h->cfg.tap_attn = tap_attn; h->cfg.tap_tfm = tap_tfm;
    int t, hh, d;
    int st = tap_geometry(h->cfg, &t, &hh, &d);
    if (st == DSIM_OK) {             // every parameter up to the new tap must have been loaded: dry walk
        Arena ar;
        Walk w{h, &ar, nullptr, 2, false};
        st = w.go(nullptr, nullptr, 0.f, 0.f, nullptr);
    }
    if (st != DSIM_OK) h->cfg = old;
    return st;
}

int dsim_unet_set_cfg_dedup(dsim_unet* h, int enable) {
    if (!h) return DSIM_ERR_INVALID;
    h->cfg_dedup = enable != 0;
    return DSIM_OK;
}

int dsim_unet_set_fusion(dsim_unet* h, int mask) {
    if (!h || (mask & ~DSIM_FUSE_ALL)) return DSIM_ERR_INVALID;
    h->fusion = mask;
    return DSIM_OK;
}

int dsim_unet_set_sample_size(dsim_unet* h, int side) {
    if (!h || side < 2) return DSIM_ERR_INVALID;          // any side: odd levels take the ceil-div / explicit-size path
    h->cfg.sample_size = side;
    return DSIM_OK;
}

int dsim_unet_qkv(dsim_unet* h, const float* latents, const float* noise, float sqrt_abar, float sqrt_1m_abar,
                  const float* ctx, int n_images, void* q, void* k, void* v, void* workspace, size_t workspace_bytes,
                  void* stream) {
    if (!h || !latents || !noise || !ctx || !q || !k || !v || !workspace || n_images < 1) return DSIM_ERR_INVALID;
    if (!h->finalized || h->timestep < 0) return DSIM_ERR_STATE;
    Arena ar;
    ar.dry = false;
    // align the arena base to 256 B
    const uintptr_t b0 = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    const size_t lost = b0 - (uintptr_t)workspace;
    if (workspace_bytes < lost) return DSIM_ERR_WORKSPACE;
    ar.base = (char*)b0;
    ar.cap = workspace_bytes - lost;
    {   // refuse up front instead of failing mid-graph
        Arena plan;
        Walk pw{h, &plan, nullptr, 2 * n_images, false};
        CK(pw.go(nullptr, nullptr, 0.f, 0.f, nullptr));
        if (plan.peak > ar.cap) return DSIM_ERR_WORKSPACE;
    }
    Walk w{h, &ar, (hipStream_t)stream, 2 * n_images, true};
    w.q_out = q; w.k_out = k; w.v_out = v;
    CK(w.go(latents, noise, sqrt_abar, sqrt_1m_abar, ctx));
    if (ar.overflow) return DSIM_ERR_WORKSPACE;
    return w.tapped ? DSIM_OK : DSIM_ERR_INVALID;
}

int dsim_unet_profile(dsim_unet* h, int enable) {
    if (!h) return DSIM_ERR_INVALID;
    for (auto& r : h->prof) {
        if (r.e0) (void)hipEventDestroy(r.e0);
        if (r.e1) (void)hipEventDestroy(r.e1);
    }
    h->prof.clear();
    h->profiling = enable != 0;
    return DSIM_OK;
}

int dsim_unet_profile_count(const dsim_unet* h) { return h ? (int)h->prof.size() : 0; }

int dsim_unet_profile_get(dsim_unet* h, int i, char* name, int name_cap, double* flops, double* bytes, double* ms) {
    if (!h || i < 0 || i >= (int)h->prof.size() || !name || name_cap < 2 || !flops || !bytes || !ms)
        return DSIM_ERR_INVALID;
    ProfRec& r = h->prof[i];
    if (r.e0 && r.e1) {
        DSIM_HIP_CHECK(hipEventSynchronize(r.e1));
        DSIM_HIP_CHECK(hipEventElapsedTime(&r.ms, r.e0, r.e1));
    }
    strncpy(name, r.name.c_str(), (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
    *flops = r.flops; *bytes = r.bytes; *ms = (double)r.ms;
    return DSIM_OK;
}

size_t dsim_pair_score_workspace_bytes(int n_pairs, int B, int H, int N, int D) {
    return pair_score_scratch_bytes(n_pairs, B, H, N, D) + 256;
}

int dsim_pair_score(const void* q, const void* k, const void* v, const int32_t* idx_a, const int32_t* idx_b, int n_pairs,
                    int B, int H, int N, int D, int dtype, int similarity, float* out_scores, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (!q || !k || !v || !idx_a || !idx_b || !out_scores || !workspace) return DSIM_ERR_INVALID;
    if (similarity != 0 && similarity != 1) return DSIM_ERR_INVALID;
    const uintptr_t b0 = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    const size_t lost = b0 - (uintptr_t)workspace;
    if (workspace_bytes < lost) return DSIM_ERR_WORKSPACE;
    return launch_pair_score(q, k, v, idx_a, idx_b, n_pairs, B, H, N, D, dtype, similarity, out_scores, (void*)b0,
                             workspace_bytes - lost, (hipStream_t)stream);
}

int dsim_pair_score_status(const void* q, const void* k, const void* v, const int32_t* idx_a, const int32_t* idx_b,
                           int n_pairs, int B, int H, int N, int D, int dtype, int similarity, float* out_scores,
                           int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !k || !v || !idx_a || !idx_b || !out_scores || !status || !workspace) return DSIM_ERR_INVALID;
    if (similarity != 0 && similarity != 1) return DSIM_ERR_INVALID;
    const uintptr_t b0 = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    const size_t lost = b0 - (uintptr_t)workspace;
    if (workspace_bytes < lost) return DSIM_ERR_WORKSPACE;
    return launch_pair_score(q, k, v, idx_a, idx_b, n_pairs, B, H, N, D, dtype, similarity, out_scores, (void*)b0,
                             workspace_bytes - lost, (hipStream_t)stream, status);
}

// ---- single-operator entry points (tests / micro-benchmarks; these allocate and synchronise) ----
namespace {
struct Tmp {
    std::vector<void*> v;
    ~Tmp() { for (void* p : v) (void)hipFree(p); }
    void* get(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        v.push_back(p);
        return p;
    }
};
}  // namespace

int dsim_op_linear(const void* x, const float* w, const float* bias, const void* residual, void* out, int M, int N,
                   int K, int dtype, int geglu, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    Tmp t;
    const int NW = geglu ? 2 * N : N;
    void* wp = t.get((size_t)NW * K * dtype_size(dtype));
    float* bp = bias ? (float*)t.get((size_t)NW * 4) : nullptr;
    void* zp = t.get(256);
    if (!wp || !zp || (bias && !bp)) return DSIM_ERR_HIP;
    DSIM_HIP_CHECK(hipMemsetAsync(zp, 0, 256, s));
    const int gblk = geglu ? geglu_block_rows(NW) : 0;
    CK(pack_linear(w, DSIM_F32, wp, dtype, NW, K, gblk, s));
    if (bias) CK(pack_vector(bias, DSIM_F32, bp, NW, gblk, s));
    GemmArgs g;
    g.A0 = x; g.C0 = K; g.mode = GEMM_LINEAR; g.M = M; g.N = NW; g.K = K; g.W = wp; g.bias = bp;
    g.epi = geglu ? EPI_GEGLU : (residual ? EPI_RESIDUAL : EPI_NONE);
    if (geglu) g.geglu_blk = gblk;
    g.residual = residual; g.out = out; g.ldo = N; g.zero_page = zp;
    CK(launch_gemm(g, dtype, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

int dsim_op_conv3x3(const void* x, const float* w, const float* bias, const void* residual, void* out, int B, int H,
                    int W, int Cin, int Cout, int stride, int upsample, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    Tmp t;
    void* wp = t.get((size_t)Cout * 9 * Cin * dtype_size(dtype));
    void* zp = t.get(256);
    if (!wp || !zp) return DSIM_ERR_HIP;
    DSIM_HIP_CHECK(hipMemsetAsync(zp, 0, 256, s));
    CK(pack_conv3(w, DSIM_F32, wp, dtype, Cout, Cin, s));
    GemmArgs g;
    g.A0 = x; g.C0 = Cin; g.mode = GEMM_CONV3; g.Hin = H; g.Win = W;
    g.Hout = upsample ? 2 * H : (stride == 2 ? (H + 1) / 2 : H);      // stride 2, padding 1: ceil(H / 2), as the executor
    g.Wout = upsample ? 2 * W : (stride == 2 ? (W + 1) / 2 : W);
    g.stride = stride; g.ups = upsample ? 1 : 0;
    g.M = B * g.Hout * g.Wout; g.N = Cout; g.K = 9 * Cin; g.W = wp; g.bias = bias;
    g.epi = residual ? EPI_RESIDUAL : EPI_NONE; g.residual = residual; g.out = out; g.ldo = Cout; g.zero_page = zp;
    CK(launch_gemm(g, dtype, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

int dsim_op_groupnorm(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta, void* out,
                      int B, int HW, int groups, float eps, int silu, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    Tmp t;
    void* sc = t.get(groupnorm_scratch_bytes(B, groups));
    if (!sc) return DSIM_ERR_HIP;
    CK(launch_groupnorm(x0, C0, x1, C1, gamma, beta, out, B, HW, groups, eps, silu, dtype, sc, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

int dsim_op_layernorm(const void* x, const float* gamma, const float* beta, void* out, int M, int C, float eps, int dtype,
                      void* stream) {
    hipStream_t s = (hipStream_t)stream;
    CK(launch_layernorm(x, gamma, beta, out, M, C, eps, dtype, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

int dsim_op_ff_fused(const void* x, const float* ln_gamma, const float* ln_beta, const float* w1, const float* b1,
                     const float* w2, const float* b2, void* out, int M, int C, float eps, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t sb = ff_stream_bytes(C);
    if (!sb || !x || !out || !w1 || !b1 || !w2 || !b2 || !ln_gamma || !ln_beta) return DSIM_ERR_INVALID;
    Tmp t;
    void* w1p = t.get((size_t)8 * C * C * 2);
    float* b1p = (float*)t.get((size_t)8 * C * 4);
    void* w2p = t.get((size_t)4 * C * C * 2);
    void* st = t.get(sb);
    if (!w1p || !b1p || !w2p || !st) return DSIM_ERR_HIP;
    CK(pack_linear(w1, DSIM_F32, w1p, DSIM_BF16, 8 * C, C, 32, s));
    CK(pack_vector(b1, DSIM_F32, b1p, 8 * C, 32, s));
    CK(pack_linear(w2, DSIM_F32, w2p, DSIM_BF16, C, 4 * C, 0, s));
    CK(pack_ff_stream(w1p, w2p, st, C, s));
    FFArgs a;
    a.x = x; a.out = out; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.stream = st; a.b1 = b1p; a.b2 = b2; a.M = M; a.C = C; a.eps = eps;
    CK(launch_ff_fused(a, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

int dsim_op_ln_linear(const void* x, const float* ln_gamma, const float* ln_beta, const float* w, void* out, int M, int C, int N,
                      float eps, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t sb = rowlin_stream_bytes(C, N);
    if (!sb || !x || !out || !w || !ln_gamma != !ln_beta) return DSIM_ERR_INVALID;
    Tmp t;
    void* wp = t.get((size_t)N * C * 2);
    void* st = t.get(sb);
    if (!wp || !st) return DSIM_ERR_HIP;
    CK(pack_linear(w, DSIM_F32, wp, DSIM_BF16, N, C, 0, s));
    CK(pack_rowlin_stream(wp, st, C, N, s));
    RowLinArgs a;
    a.x = x; a.out = out; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.stream = st; a.M = M; a.C = C; a.N = N; a.eps = eps;
    CK(launch_rowlin(a, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

int dsim_op_attention(const void* q, int ldq, const void* k, const void* v, int ldk, void* out, int ldo, int B, int Bkv,
                      int H, int Nq, int Nk, int D, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    AttnArgs a;
    a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldk = ldk; a.out = out; a.ldo = ldo;
    a.B = B; a.Bkv = Bkv; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D;
    CK(launch_attention(a, dtype, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

int dsim_op_attention_fp8(const void* q, int ldq, const void* k, const void* v, int ldk, void* out, int ldo, int B, int Bkv,
                          int H, int Nq, int Nk, int D, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    AttnArgs a;
    a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldk = ldk; a.out = out; a.ldo = ldo;
    a.B = B; a.Bkv = Bkv; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D;
    CK(launch_attention_fp8(a, s));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    return DSIM_OK;
}

}  // extern "C"
