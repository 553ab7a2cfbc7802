// Shared host-side plumbing of the engine handles (U-Net, VAE): borrowed raw parameters, packed
// device copies, the caller-provided workspace arena and the per-launch profiling record.
#pragma once
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace dsim {

struct RawW { const void* p; int dtype; std::vector<int64_t> shape; };

struct Packed { void* p = nullptr; size_t bytes = 0; int rows = 0, cols = 0; };

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = true;
    bool overflow = false;
    void* alloc(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        if (off > peak) peak = off;
        if (!dry && off > cap) { overflow = true; return nullptr; }
        return dry ? (void*)(uintptr_t)(a + 256) : (void*)(base + a);   // non-null dummy when planning
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

struct Act { void* p = nullptr; int C = 0, H = 0, W = 0; };

// per-launch record of a profiled forward (dsim_*_profile_*): kernel family, algorithmic work
// and the HIP-event bracket on the launch stream
struct ProfRec {
    std::string name;
    double flops = 0, bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
};

#define CK(expr)                         \
    do {                                 \
        int _s = (expr);                 \
        if (_s != DSIM_OK) return _s;    \
    } while (0)

static inline bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// parameters of one model: raw (borrowed until finalize) and packed (owned device buffers)
struct WeightStore {
    int dt = DSIM_BF16;
    bool finalized = false;
    std::map<std::string, RawW> raw;
    std::map<std::string, Packed> pk;
    std::vector<void*> owned;
    void* zero_page = nullptr;
    std::string err_key;
    bool profiling = false;
    std::vector<ProfRec> prof;

    int dalloc(size_t bytes, void** out) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return DSIM_ERR_HIP;
        owned.push_back(p);
        *out = p;
        return DSIM_OK;
    }
    const Packed* find(const std::string& k) {
        auto it = pk.find(k);
        if (it == pk.end()) { err_key = k; return nullptr; }
        return &it->second;
    }
    void free_all() {
        for (void* p : owned) (void)hipFree(p);
        owned.clear();
    }
    int add_raw(const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim) {
        if (!key || !dev_ptr || !shape || ndim < 1 || ndim > 4) return DSIM_ERR_INVALID;
        if (dtype != DSIM_F32 && dtype != DSIM_BF16 && dtype != DSIM_F16) return DSIM_ERR_INVALID;
        if (finalized) return DSIM_ERR_STATE;
        RawW w;
        w.p = dev_ptr; w.dtype = dtype; w.shape.assign(shape, shape + ndim);
        raw[key] = w;
        return DSIM_OK;
    }
    void clear_profile() {
        for (auto& r : prof) {
            if (r.e0) (void)hipEventDestroy(r.e0);
            if (r.e1) (void)hipEventDestroy(r.e1);
        }
        prof.clear();
    }
};

// profile family of a GEMM launch = the kernel symbol launch_gemm will pick for it (one family per symbol), + its algorithmic work
static inline std::string gemm_family(const GemmArgs& g, int dt, double* flops, double* bytes) {
    int bm, bn;
    gemm_launch_tile(g, dt, &bm, &bn);
    const bool skinny = dt != DSIM_F32 && !(g.wb_rows && g.wb_rows != g.M) && !g.force_big && !g.gn_part && gemm_skinny_applies(g);      // the small-batch kernel (gemm_skinny.hip)
    // 256-row 16-bit conv tiles on power-of-two output maps run the CONV3P instantiation (gemm.hip launch_typed)
    const int hwo = g.Hout * g.Wout;
    const bool conv_p2 = g.mode == GEMM_CONV3 && !skinny && dt != DSIM_F32 && (bm == 256 || bm == 512) && g.Wout > 0 &&
                         !(g.Wout & (g.Wout - 1)) && !(hwo & (hwo - 1));
    if (skinny) gemm_skinny_tile(g, &bm, &bn);
    const char* dtn = dt == DSIM_F32 ? "f32" : (dt == DSIM_F16 ? "f16" : "bf16");
    const double e = (double)dtype_size(dt);
    const double outc = g.epi == EPI_GEGLU ? g.N / 2 : g.N;
    *flops = 2.0 * g.M * (double)g.N * g.K;
    *bytes = e * ((double)g.M * (g.mode == GEMM_CONV3 ? g.C0 : g.K) + (double)g.N * g.K * (g.wb_rows ? g.M / g.wb_rows : 1) +
                  (double)g.M * outc * (g.residual ? 2 : 1));
    return std::string(skinny ? "gemm_small_" : "gemm_") + dtn + "_" + std::to_string(bm) + "x" + std::to_string(bn) +
           (g.mode == GEMM_CONV3 ? (conv_p2 ? "_conv3p" : "_conv3") : "_linear") +
           (g.epi == EPI_GEGLU ? "_geglu" : (g.epi == EPI_RESIDUAL ? "_res" : "")) + (g.gn_part ? "_gn" : "") +      // _gn: the statistics epilogue kinds
           "|M" + std::to_string(g.M) + " N" + std::to_string(g.N) + " K" + std::to_string(g.K);
}

// HIP-event bracket of one launch of a profiled forward (the executors' pbegin / pend)
static inline void prof_begin(WeightStore* h, hipStream_t s, const std::string& name, double flops, double bytes) {
    ProfRec r;
    r.name = name; r.flops = flops; r.bytes = bytes;
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, s);
    h->prof.push_back(r);
}
static inline void prof_end(WeightStore* h, hipStream_t s) { (void)hipEventRecord(h->prof.back().e1, s); }
static inline int prof_get(WeightStore* h, int i, char* name, int name_cap, double* flops, double* bytes, double* ms) {
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

// Repack every raw parameter into the engine's layouts.
// ---- weight packing (finalize) ---------------------------------------------------------------
static inline int pack_all(WeightStore* h, hipStream_t s) {
    const int dt = h->dt;
    const size_t es = dtype_size(dt);
    for (auto& kv : h->raw) {
        const std::string& key = kv.first;
        const RawW& w = kv.second;
        Packed P;
        const bool keep_f32 = key == "pos_embed" || ends_with(key, "x_embedder.proj.weight") ||
                              ends_with(key, "embedding_table.weight") || key.rfind("t_embedder.", 0) == 0 ||
                              ends_with(key, "adaLN_modulation.1.weight");
        if (keep_f32 && w.shape.size() >= 2) {
            // small DiT conditioning tensors consumed by f32 GEMV / lookup / patch-embed kernels: flat f32 copy
            int64_t n = 1;
            for (auto d : w.shape) n *= d;
            CK(h->dalloc((size_t)n * 4, &P.p));
            P.rows = (int)w.shape[0]; P.cols = (int)(n / w.shape[0]); P.bytes = (size_t)n * 4;
            CK(pack_vector(w.p, w.dtype, (float*)P.p, (int)n, 0, s));
            h->pk[key] = P;
        } else if (w.shape.size() == 1) {
            const int n = (int)w.shape[0];
            const int geglu = ends_with(key, "ff.net.0.proj.bias") ? geglu_block_rows(n) : 0;
            CK(h->dalloc((size_t)n * 4, &P.p));
            P.rows = n; P.cols = 1; P.bytes = (size_t)n * 4;
            CK(pack_vector(w.p, w.dtype, (float*)P.p, n, geglu, s));
            h->pk[key] = P;
        } else if (w.shape.size() == 4 && w.shape[2] == 3) {
            const int co = (int)w.shape[0], ci = (int)w.shape[1];
            if (ends_with(key, "conv_in.weight")) {
                CK(h->dalloc((size_t)co * ci * 9 * 4, &P.p));
                P.rows = 9 * ci; P.cols = co;
                CK(pack_conv_in(w.p, w.dtype, (float*)P.p, co, ci, s));
            } else {
                CK(h->dalloc((size_t)co * ci * 9 * es, &P.p));
                P.rows = co; P.cols = 9 * ci;
                CK(pack_conv3(w.p, w.dtype, P.p, dt, co, ci, s));
            }
            h->pk[key] = P;
        } else if (w.shape.size() == 2 || (w.shape.size() == 4 && w.shape[2] == 1)) {
            const int n = (int)w.shape[0], k = (int)w.shape[1];
            const bool is_q1 = ends_with(key, "attn1.to_q.weight"), is_k1 = ends_with(key, "attn1.to_k.weight"),
                       is_v1 = ends_with(key, "attn1.to_v.weight");
            const bool is_k2 = ends_with(key, "attn2.to_k.weight"), is_v2 = ends_with(key, "attn2.to_v.weight");
            if (is_q1 || is_k1 || is_v1) {
                // fused [3C][C] = [to_q ; to_k ; to_v]
                const std::string fk = key.substr(0, key.size() - strlen("to_q.weight")) + "qkv";
                Packed& F = h->pk[fk];
                if (!F.p) { CK(h->dalloc((size_t)3 * n * k * es, &F.p)); F.rows = 3 * n; F.cols = k; }
                const int slot = is_q1 ? 0 : (is_k1 ? 1 : 2);
                CK(pack_linear(w.p, w.dtype, (char*)F.p + (size_t)slot * n * k * es, dt, n, k, 0, s));
            } else if (is_k2 || is_v2) {
                const std::string fk = key.substr(0, key.size() - strlen("to_k.weight")) + "kv";
                Packed& F = h->pk[fk];
                if (!F.p) { CK(h->dalloc((size_t)2 * n * k * es, &F.p)); F.rows = 2 * n; F.cols = k; }
                CK(pack_linear(w.p, w.dtype, (char*)F.p + (size_t)(is_k2 ? 0 : 1) * n * k * es, dt, n, k, 0, s));
            } else if (ends_with(key, "time_emb_proj.weight") || key.rfind("time_embedding.", 0) == 0 ||
                       key.rfind("add_embedding.", 0) == 0) {
                CK(h->dalloc((size_t)n * k * 4, &P.p));        // kept f32: consumed by the GEMV
                P.rows = n; P.cols = k;
                CK(pack_linear(w.p, w.dtype, P.p, DSIM_F32, n, k, 0, s));
                h->pk[key] = P;
            } else {
                const int geglu = ends_with(key, "ff.net.0.proj.weight") ? geglu_block_rows(n) : 0;
                CK(h->dalloc((size_t)n * k * es, &P.p));
                P.rows = n; P.cols = k;
                CK(pack_linear(w.p, w.dtype, P.p, dt, n, k, geglu, s));
                h->pk[key] = P;
            }
        } else {
            return DSIM_ERR_INVALID;
        }
    }
    return DSIM_OK;
}

}  // namespace dsim
