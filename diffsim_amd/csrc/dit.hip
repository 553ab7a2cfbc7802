// DiT-XL/2 executor (tokens -> tapped attention q/k/v) and its C ABI -- SURVEY.md section 8a row a11.
//
// Replaces, for the reference's DiT scorer (diffsim/diffsim_dit.py:74-142), the call
// `diffusion.p_sample(model, latents, t, model_kwargs=dict(y=[1,1000]))`, which through _WrappedModel
// (DiT/diffusion/respace.py:117-129) runs `DiT.forward(x, timestep_map[t], y)` (DiT/modelsdit.py:235-250)
// and returns right after the model call (DiT/diffusion/gaussian_diffusion.py:279-280); the forward
// pre-hook on `model.blocks[L].attn` (diffsim_dit.py:19-26,100) stores q,k,v = split(qkv(input)).
// Graph per block (adaLN-Zero, DiT/modelsdit.py:103-124):
//     x += gate_msa * proj(attn(LN(x) * (1 + scale_msa) + shift_msa))
//     x += gate_mlp * fc2(gelu_tanh(fc1(LN(x) * (1 + scale_mlp) + shift_mlp)))
// The batch quirk is reproduced: x has batch 1 but c = t_emb + y_emb has batch 2 (class 1, null class),
// so every image is carried as two rows-blocks that share the patch embedding and differ in modulation.
// c is constant for a run, so all 6*depth modulation vectors are precomputed by dsim_dit_set_conditioning.
// Kernels: patch embedding (K = 16: direct), modulated LayerNorm (norm.hip), the MFMA GEMM with bias /
// tanh-GELU / adaLN-gate+residual epilogues (gemm.hip), flash attention with D = 72 (attention.hip).
#include <string>

#include "common.h"
#include "store.h"

using namespace dsim;

struct dsim_dit : WeightStore {
    dsim_dit_cfg cfg;
    float* cvec = nullptr;      // [2][D]   c = t_emb + y_emb for the two halves
    float* mod = nullptr;       // [depth][6][2][D]  shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    float* scratch = nullptr;   // freq[freq_dim] + h1[D] + temb[D] + m6[6D]
    bool cond_set = false;
    int attn_mode = 0;          // 0: compute dtype, 1: fp8 e4m3 MFMAs (dsim_dit_set_attention)
};

namespace {

// x_t = sa*z + sb*eps (NCHW f32), non-overlapping p x p patches -> token rows, + bias + pos_embed, written for
// both halves.  grid (T/TOK, n_img), 256 threads; w is [D][Cin*p*p] f32.
constexpr int PE_TOK = 8;
template <typename T>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ lat, const float* __restrict__ noise,
                                                          float sa, float sb, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ pos,
                                                          T* __restrict__ out, int Cin, int S, int p, int D) {
    extern __shared__ float patch[];            // [PE_TOK][K]
    const int K = Cin * p * p, g = S / p, Ttok = g * g;
    const int img = blockIdx.y, t0 = blockIdx.x * PE_TOK;
    for (int i = threadIdx.x; i < PE_TOK * K; i += 256) {
        const int tk = i / K, k = i - tk * K;
        const int c = k / (p * p), r = k - c * p * p, py = r / p, px = r - py * p;
        const int tok = t0 + tk;
        float v = 0.f;
        if (tok < Ttok) {
            const int ty = tok / g, tx = tok - ty * g;
            const size_t o = (((size_t)img * Cin + c) * S + ty * p + py) * S + tx * p + px;
            v = sa * lat[o] + sb * noise[o];
        }
        patch[i] = v;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc[PE_TOK];
#pragma unroll
        for (int tk = 0; tk < PE_TOK; ++tk) acc[tk] = bias[d];
        for (int k = 0; k < K; ++k) {
            const float wv = w[(size_t)d * K + k];
#pragma unroll
            for (int tk = 0; tk < PE_TOK; ++tk) acc[tk] = fmaf(patch[tk * K + k], wv, acc[tk]);
        }
#pragma unroll
        for (int tk = 0; tk < PE_TOK; ++tk) {
            const int tok = t0 + tk;
            if (tok < Ttok) {
                const T v = (T)(acc[tk] + pos[(size_t)tok * D + d]);
                out[((size_t)(img * 2 + 0) * Ttok + tok) * D + d] = v;
                out[((size_t)(img * 2 + 1) * Ttok + tok) * D + d] = v;
            }
        }
    }
}

__global__ void add_table_row_kernel(const float* a, const float* table, int row, float* out, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) out[i] = a[i] + table[(size_t)row * D + i];
}

struct DWalk {
    dsim_dit* h;
    Arena* ar;
    hipStream_t s;
    int n;                  // images
    bool run;
    void *q_out = nullptr, *k_out = nullptr, *v_out = nullptr;
    bool tapped = false;

    size_t es() const { return dtype_size(h->dt); }
    void* alloc_act(size_t elems) { return ar->alloc(elems * es()); }
#define DGET(var, key)                                   \
    const Packed* var = h->find(key);                    \
    if (!var) return DSIM_ERR_MISSING_WEIGHT;
    const float* modv(int blk, int chunk) const { return h->mod + (((size_t)blk * 6 + chunk) * 2) * h->cfg.hidden_size; }

    int linear(const void* a, int K, const void* w, const float* bias, void* out, int M, int N, int act, const float* gate,
               const void* residual, int T) {
        GemmArgs g;
        g.A0 = a; g.C0 = K; g.mode = GEMM_LINEAR; g.M = M; g.N = N; g.K = K; g.W = w; g.bias = bias; g.act = act;
        if (gate) { g.gate = gate; g.gate2 = gate + h->cfg.hidden_size; g.rows_per_batch = T; }
        g.epi = residual ? EPI_RESIDUAL : EPI_NONE; g.residual = residual; g.out = out; g.ldo = N;
        g.zero_page = h->zero_page;
        if (!run) return DSIM_OK;
        if (h->profiling) {
            int bm, bn;
            gemm_launch_tile(g, h->dt, &bm, &bn);
            const bool slow = act != 0 || gate != nullptr;      // the tanh-GELU / adaLN-gate epilogue template (gemm.hip EK_SLOW)
            pbegin(std::string("gemm_") + dtn() + "_" + std::to_string(bm) + "x" + std::to_string(bn) + "_linear" +
                       (slow ? ((act && !gate && !residual) ? "_act" : "_dit") : (residual ? "_res" : "")) + "|M" + std::to_string(M) + " N" + std::to_string(N) + " K" + std::to_string(K),
                   2.0 * M * (double)N * K, (double)es() * ((double)M * K + (double)N * K + (double)M * N * (residual ? 2 : 1)));
        }
        const int st = launch_gemm(g, h->dt, s);
        pend();
        return st;
    }
    // per-launch HIP-event brackets of a profiled forward (same record format as the U-Net executor's)
    const char* dtn() const { return h->dt == DSIM_F32 ? "f32" : (h->dt == DSIM_F16 ? "f16" : "bf16"); }
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
    int lnmod(const void* x, const float* scale2, const float* shift2, void* out, int M, int D, int T) {
        if (!run) return DSIM_OK;
        pbegin(std::string("layernorm_mod_") + dtn() + "|M" + std::to_string(M) + " C" + std::to_string(D), 0.0, 2.0 * M * (double)D * es());
        const int st = launch_layernorm_mod(x, scale2, shift2, out, M, D, T, 1e-6f, h->dt, s);
        pend();
        return st;
    }

    int go(const float* lat, const float* noise, float sa, float sb) {
        const dsim_dit_cfg& c = h->cfg;
        const int D = c.hidden_size, p = c.patch_size, S = c.input_size, g = S / p, T = g * g, H = c.num_heads;
        const int M = n * 2 * T, F = c.mlp_ratio * D;
        DGET(pw, "x_embedder.proj.weight"); DGET(pb, "x_embedder.proj.bias"); DGET(pos, "pos_embed");
        void* x = alloc_act((size_t)M * D);
        if (run) {
            const int K = c.in_channels * p * p;
            const dim3 grid((T + PE_TOK - 1) / PE_TOK, n);
            if (h->dt == DSIM_BF16)
                hipLaunchKernelGGL(patch_embed_kernel<bf16_t>, grid, dim3(256), PE_TOK * K * sizeof(float), s, lat, noise, sa, sb,
                                   (const float*)pw->p, (const float*)pb->p, (const float*)pos->p, (bf16_t*)x, c.in_channels, S, p, D);
            else if (h->dt == DSIM_F16)
                hipLaunchKernelGGL(patch_embed_kernel<f16_t>, grid, dim3(256), PE_TOK * K * sizeof(float), s, lat, noise, sa, sb,
                                   (const float*)pw->p, (const float*)pb->p, (const float*)pos->p, (f16_t*)x, c.in_channels, S, p, D);
            else
                hipLaunchKernelGGL(patch_embed_kernel<float>, grid, dim3(256), PE_TOK * K * sizeof(float), s, lat, noise, sa, sb,
                                   (const float*)pw->p, (const float*)pb->p, (const float*)pos->p, (float*)x, c.in_channels, S, p, D);
            DSIM_HIP_CHECK(hipGetLastError());
        }
        void* nb = alloc_act((size_t)M * D);
        void* big = alloc_act((size_t)M * (F > 3 * D ? F : 3 * D));
        void* ab = alloc_act((size_t)M * D);
        for (int blk = 0; blk <= c.tap_layer; ++blk) {
            const std::string b = "blocks." + std::to_string(blk) + ".";
            DGET(qw, b + "attn.qkv.weight"); DGET(qb, b + "attn.qkv.bias");
            CK(lnmod(x, modv(blk, 1), modv(blk, 0), nb, M, D, T));
            if (blk == c.tap_layer) {
                // the pre-hook's input is the modulated norm1 output; q/k/v = row blocks of the fused qkv Linear
                for (int j = 0; j < 3; ++j) {
                    void* dst = j == 0 ? q_out : (j == 1 ? k_out : v_out);
                    CK(linear(nb, D, (char*)qw->p + (size_t)j * D * D * es(), (const float*)qb->p + (size_t)j * D, dst, M, D, 0,
                              nullptr, nullptr, T));
                }
                tapped = true;
                return DSIM_OK;
            }
            DGET(ow, b + "attn.proj.weight"); DGET(ob, b + "attn.proj.bias");
            DGET(f1w, b + "mlp.fc1.weight"); DGET(f1b, b + "mlp.fc1.bias");
            DGET(f2w, b + "mlp.fc2.weight"); DGET(f2b, b + "mlp.fc2.bias");
            CK(linear(nb, D, qw->p, (const float*)qb->p, big, M, 3 * D, 0, nullptr, nullptr, T));
            if (run) {
                AttnArgs a;
                a.q = big; a.ldq = 3 * D;
                a.k = (char*)big + (size_t)D * es(); a.v = (char*)big + (size_t)2 * D * es(); a.ldk = 3 * D;
                a.out = ab; a.ldo = D; a.B = n * 2; a.Bkv = n * 2; a.H = H; a.Nq = T; a.Nk = T; a.D = D / H;
                pbegin((h->attn_mode == 1 ? std::string("attention_fp8_d") + std::to_string(a.D)
                                           : std::string("attention_") + dtn() + "_d" + std::to_string(a.D) + attention_kernel_kind(a, h->dt)) + "|B" +
                           std::to_string(a.B) + " H" + std::to_string(H) + " Nq" + std::to_string(T) + " Nk" + std::to_string(T),
                       4.0 * a.B * H * (double)T * T * a.D, (double)es() * a.B * H * a.D * 4.0 * T);
                const int st = h->attn_mode == 1 ? launch_attention_fp8(a, s) : launch_attention(a, h->dt, s);
                pend();
                CK(st);
            }
            CK(linear(ab, D, ow->p, (const float*)ob->p, x, M, D, 0, modv(blk, 2), x, T));
            CK(lnmod(x, modv(blk, 4), modv(blk, 3), nb, M, D, T));
            CK(linear(nb, D, f1w->p, (const float*)f1b->p, big, M, F, 1, nullptr, nullptr, T));
            CK(linear(big, F, f2w->p, (const float*)f2b->p, x, M, D, 0, modv(blk, 5), x, T));
        }
        return DSIM_ERR_INVALID;
    }
};

}  // namespace

extern "C" {

int dsim_dit_create(const dsim_dit_cfg* cfg, dsim_dit** out) {
    if (!cfg || !out) return DSIM_ERR_INVALID;
    if (cfg->compute_dtype != DSIM_F32 && cfg->compute_dtype != DSIM_BF16 && cfg->compute_dtype != DSIM_F16) return DSIM_ERR_INVALID;
    if (cfg->tap_layer < 0 || cfg->tap_layer >= cfg->depth || cfg->hidden_size % cfg->num_heads ||
        cfg->input_size % cfg->patch_size)
        return DSIM_ERR_INVALID;
    if (dsim_device_count() < 1) return DSIM_ERR_NO_DEVICE;
    dsim_dit* h = new dsim_dit();
    h->cfg = *cfg;
    h->dt = cfg->compute_dtype;
    if (h->dalloc(256, &h->zero_page) != DSIM_OK || hipMemset(h->zero_page, 0, 256) != hipSuccess) {
        dsim_dit_destroy(h);
        return DSIM_ERR_HIP;
    }
    *out = h;
    return DSIM_OK;
}

void dsim_dit_destroy(dsim_dit* h) {
    if (!h) return;
    h->free_all();
    delete h;
}

int dsim_dit_load_weight(dsim_dit* h, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim) {
    if (!h) return DSIM_ERR_INVALID;
    return h->add_raw(key, dev_ptr, dtype, shape, ndim);
}

int dsim_dit_finalize(dsim_dit* h, void* stream) {
    if (!h) return DSIM_ERR_INVALID;
    if (h->finalized) return DSIM_ERR_STATE;
    hipStream_t s = (hipStream_t)stream;
    CK(pack_all(h, s));
    const int D = h->cfg.hidden_size;
    CK(h->dalloc((size_t)2 * D * 4, (void**)&h->cvec));
    CK(h->dalloc((size_t)h->cfg.depth * 6 * 2 * D * 4, (void**)&h->mod));
    CK(h->dalloc((size_t)(h->cfg.freq_dim + 8 * D + 64) * 4, (void**)&h->scratch));
    DSIM_HIP_CHECK(hipStreamSynchronize(s));
    h->raw.clear();
    h->finalized = true;
    Arena ar;
    DWalk w{h, &ar, s, 1, false};
    const int st = w.go(nullptr, nullptr, 0.f, 0.f);
    if (st != DSIM_OK) { h->finalized = false; return st; }
    return DSIM_OK;
}

int dsim_dit_set_conditioning(dsim_dit* h, int t_model, int y0, int y1, void* stream) {
    if (!h || t_model < 0) return DSIM_ERR_INVALID;
    if (!h->finalized) return DSIM_ERR_STATE;
    const dsim_dit_cfg& c = h->cfg;
    if (y0 < 0 || y0 > c.num_classes || y1 < 0 || y1 > c.num_classes) return DSIM_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int D = c.hidden_size, fd = c.freq_dim;
    float* freq = h->scratch;          // [fd]
    float* h1 = freq + fd;             // [D]
    float* temb = h1 + D;              // [D]
    float* m6 = temb + D;              // [6D]
    const Packed* w0 = h->find("t_embedder.mlp.0.weight");
    const Packed* b0 = h->find("t_embedder.mlp.0.bias");
    const Packed* w2 = h->find("t_embedder.mlp.2.weight");
    const Packed* b2 = h->find("t_embedder.mlp.2.bias");
    const Packed* tab = h->find("y_embedder.embedding_table.weight");
    if (!w0 || !b0 || !w2 || !b2 || !tab) return DSIM_ERR_MISSING_WEIGHT;
    CK(timestep_sincos(freq, fd, t_model, s));                                   // [cos | sin], DiT/modelsdit.py:42-60
    CK(gemv_f32(w0->p, DSIM_F32, b0->p, DSIM_F32, freq, h1, D, fd, 0, s));
    CK(gemv_f32(w2->p, DSIM_F32, b2->p, DSIM_F32, h1, temb, D, D, 1, s));
    const int ys[2] = {y0, y1};
    for (int half = 0; half < 2; ++half) {
        float* cv = h->cvec + (size_t)half * D;
        hipLaunchKernelGGL(add_table_row_kernel, dim3((D + 255) / 256), dim3(256), 0, s, temb, (const float*)tab->p, ys[half], cv, D);
        // every block whose weights are loaded, not only those up to the current tap: dsim_dit_set_tap may move it later
        for (int blk = 0; blk < c.depth; ++blk) {
            const std::string b = "blocks." + std::to_string(blk) + ".adaLN_modulation.1.";
            const Packed* aw = h->find(b + "weight");
            const Packed* ab = h->find(b + "bias");
            if (!aw || !ab) {
                if (blk <= c.tap_layer) return DSIM_ERR_MISSING_WEIGHT;
                break;
            }
            CK(gemv_f32(aw->p, DSIM_F32, ab->p, DSIM_F32, cv, m6, 6 * D, D, 1, s));        // Linear(SiLU(c))
            for (int j = 0; j < 6; ++j)
                DSIM_HIP_CHECK(hipMemcpyAsync(h->mod + ((((size_t)blk * 6 + j) * 2) + half) * D, m6 + (size_t)j * D, (size_t)D * 4,
                                              hipMemcpyDeviceToDevice, s));
        }
    }
    DSIM_HIP_CHECK(hipGetLastError());
    h->cond_set = true;
    return DSIM_OK;
}

int dsim_dit_set_tap(dsim_dit* h, int tap_layer) {
    if (!h || tap_layer < 0 || tap_layer >= h->cfg.depth) return DSIM_ERR_INVALID;
    if (!h->finalized) return DSIM_ERR_STATE;
    const int old = h->cfg.tap_layer;
    if (tap_layer == old) return DSIM_OK;
    h->cfg.tap_layer = tap_layer;
    Arena ar;                                   // every parameter up to the new tap must have been loaded: dry walk
    DWalk w{h, &ar, nullptr, 1, false};
    const int st = w.go(nullptr, nullptr, 0.f, 0.f);
    if (st != DSIM_OK) { h->cfg.tap_layer = old; return st; }
    return DSIM_OK;
}

int dsim_dit_set_attention(dsim_dit* h, int mode) {
    if (!h || (mode != 0 && mode != 1)) return DSIM_ERR_INVALID;
    const int hd = h->cfg.num_heads > 0 ? h->cfg.hidden_size / h->cfg.num_heads : 0;
    if (mode == 1 && (h->dt != DSIM_BF16 || (hd != 72 && hd != 32))) return DSIM_ERR_INVALID;
    h->attn_mode = mode;
    return DSIM_OK;
}

int dsim_dit_profile(dsim_dit* h, int enable) {
    if (!h) return DSIM_ERR_INVALID;
    h->clear_profile();
    h->profiling = enable != 0;
    return DSIM_OK;
}

int dsim_dit_profile_count(const dsim_dit* h) { return h ? (int)h->prof.size() : 0; }

int dsim_dit_profile_get(dsim_dit* h, int i, char* name, int name_cap, double* flops, double* bytes, double* ms) {
    if (!h || i < 0 || i >= (int)h->prof.size() || !name || name_cap < 2 || !flops || !bytes || !ms) return DSIM_ERR_INVALID;
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

size_t dsim_dit_workspace_bytes(const dsim_dit* hc, int n_images) {
    dsim_dit* h = const_cast<dsim_dit*>(hc);
    if (!h || !h->finalized || n_images < 1) return 0;
    Arena ar;
    DWalk w{h, &ar, nullptr, n_images, false};
    if (w.go(nullptr, nullptr, 0.f, 0.f) != DSIM_OK) return 0;
    return ar.peak + 256;
}

int dsim_dit_qkv(dsim_dit* h, const float* latents, const float* noise, float sqrt_abar, float sqrt_1m_abar, int n_images,
                 void* q, void* k, void* v, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !latents || !noise || !q || !k || !v || !workspace || n_images < 1) return DSIM_ERR_INVALID;
    if (!h->finalized || !h->cond_set) return DSIM_ERR_STATE;
    Arena ar;
    ar.dry = false;
    const uintptr_t b0 = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    const size_t lost = b0 - (uintptr_t)workspace;
    if (workspace_bytes < lost) return DSIM_ERR_WORKSPACE;
    ar.base = (char*)b0;
    ar.cap = workspace_bytes - lost;
    {
        Arena plan;
        DWalk pw{h, &plan, nullptr, n_images, false};
        CK(pw.go(nullptr, nullptr, 0.f, 0.f));
        if (plan.peak > ar.cap) return DSIM_ERR_WORKSPACE;
    }
    DWalk w{h, &ar, (hipStream_t)stream, n_images, true};
    w.q_out = q; w.k_out = k; w.v_out = v;
    CK(w.go(latents, noise, sqrt_abar, sqrt_1m_abar));
    return w.tapped && !ar.overflow ? DSIM_OK : DSIM_ERR_WORKSPACE;
}

}  // extern "C"
