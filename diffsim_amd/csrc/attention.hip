// Flash-style attention and the fused DiffSim score tail for gfx950.
//
//  * attn_kernel     : softmax(Q K^T / sqrt(D)) V for the U-Net's self/cross attention layers
//                      (replaces F.scaled_dot_product_attention inside diffusers' AttnProcessor2_0,
//                      restated at /root/reference/diffsim/hacked_attn.py:74-83).
//  * pair_tail_kernel: the DiffSim score tail -- /root/reference/diffsim/diffsim.py:177-197:
//                      O_ab = SDPA(Qa,Kb,Vb), O_aa = SDPA(Qa,Ka,Va) (and the b<->a mirror), then
//                      cosine (or mse) over the flattened (B,H,N,D) tensors.  Both attentions of a
//                      direction share Q and run in one workgroup; O never leaves registers, only
//                      three f32 partial sums per workgroup reach HBM and a second fixed-order pass
//                      folds them (no float atomics => bit-reproducible scores).
//
// Tiling: a workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows.
// S^T = K Q^T is computed with K as the MFMA A operand and Q as B ("swapped QK^T"), so a lane
// holds one query column of S^T in its accumulator registers: the row max/sum are per-lane
// reductions over registers plus one exchange between the two lane halves, and the accumulator
// is directly the B operand of O^T = V^T P^T (no LDS round trip for P).  V^T fragments come from
// the row-major V tile by ds_read_b64_tr_b16 (bf16) or ds_read_b32 (f32).
// bf16 path: v_mfma_f32_32x32x16_bf16; fp32 parity path: v_mfma_f32_32x32x2_f32 (exact f32).
#include "common.h"

namespace dsim {
namespace {

constexpr int KT = 64;   // kv rows per LDS tile

template <typename T, int DP> struct ACfg {
    static constexpr int ES = sizeof(T);
    static constexpr int VEC = 16 / ES;
    static constexpr int RS = DP * ES + 16;       // LDS row stride: odd number of 16-B slots
    static constexpr int CPR = DP / VEC;          // 16-B chunks per row
    static constexpr int NDB = DP / 32;           // 32-wide output blocks over d
    static constexpr int NKS = DP / 16;           // 16-deep k steps over d
    static constexpr int TILE = KT * RS;
    static constexpr int LDS = 2 * TILE;
};

struct FragF32 { f32x4 lo, hi; };
template <typename T> struct FragOf { typedef bf16x8 type; };
template <> struct FragOf<float> { typedef FragF32 type; };

__device__ __forceinline__ void zero_frag(bf16x8& f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16)0.0f;
}
__device__ __forceinline__ void zero_frag(FragF32& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.lo[i] = f.hi[i] = 0.f;
}
__device__ __forceinline__ void gload_frag(bf16x8& f, const bf16* p) { f = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void gload_frag(FragF32& f, const float* p) {
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 4);
}
__device__ __forceinline__ void lload_frag(bf16x8& f, const char* p) { f = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void lload_frag(FragF32& f, const char* p) {
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 16);
}
__device__ __forceinline__ void mma(const bf16x8& a, const bf16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const FragF32& a, const FragF32& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], c, 0, 0, 0);
}

// Q fragments of this wave's 32 query rows, resident in registers for the whole kv sweep.
template <typename T, int DP> struct QFrags { typename FragOf<T>::type f[ACfg<T, DP>::NKS]; };
template <typename T, int DP> struct OAcc { f32x16 b[ACfg<T, DP>::NDB]; };

template <typename T, int DP>
__device__ __forceinline__ void load_q(QFrags<T, DP>& qf, const T* qrow /*row base + h*D*/, int D, int half) {
#pragma unroll
    for (int ks = 0; ks < ACfg<T, DP>::NKS; ++ks) {
        const int d0 = 16 * ks + 8 * half;
        if (d0 < D) gload_frag(qf.f[ks], qrow + d0);
        else zero_frag(qf.f[ks]);
    }
}

// Stage one KT-row tile of K and V (rows >= Nk and columns >= D zero-filled) into LDS.
template <typename T, int DP>
__device__ __forceinline__ void stage_kv(char* lds, const T* kb, const T* vb, int ldk, int kv0, int Nk, int D, int tid) {
    typedef ACfg<T, DP> C;
    u32x4 z = {0u, 0u, 0u, 0u};
    for (int idx = tid; idx < KT * C::CPR; idx += 256) {
        const int r = idx / C::CPR, c = idx - r * C::CPR;
        const int kv = kv0 + r;
        u32x4 kvv = z, vvv = z;
        if (kv < Nk && c * C::VEC < D) {
            const size_t off = (size_t)kv * ldk + c * C::VEC;
            kvv = *reinterpret_cast<const u32x4*>(kb + off);
            vvv = *reinterpret_cast<const u32x4*>(vb + off);
        }
        *reinterpret_cast<u32x4*>(lds + r * C::RS + c * 16) = kvv;
        *reinterpret_cast<u32x4*>(lds + C::TILE + r * C::RS + c * 16) = vvv;
    }
}

// One full attention of this wave's 32 query rows against Nk keys.  On return o[db][r] holds the
// NORMALISED output O^T[d = db*32 + (r&3)+8(r>>2)+4*half][q = lane&31].  All 256 threads of the
// workgroup must call it together (it contains workgroup barriers).
template <typename T, int DP>
__device__ __forceinline__ void attend(const QFrags<T, DP>& qfr, const T* kb, const T* vb, int ldk, int Nk, int D,
                                       float scale_log2, char* lds, OAcc<T, DP>& oacc) {
    const auto& qf = qfr.f;
    auto& o = oacc.b;
    typedef ACfg<T, DP> C;
    typedef typename FragOf<T>::type Frag;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (Nk + KT - 1) / KT;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();                                   // previous tile fully consumed
        stage_kv<T, DP>(lds, kb, vb, ldk, kt * KT, Nk, D, tid);
        __syncthreads();

        // ---- S^T = K Q^T for the two 32-row kv blocks ------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[jb][r] = 0.f;
            const char* krow = lds + (jb * 32 + l31) * C::RS + half * 8 * C::ES;
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                if (ks * 16 < D) {
                    Frag kf;
                    lload_frag(kf, krow + ks * 16 * C::ES);
                    mma(kf, qf[ks], s[jb]);
                }
            }
        }
        // ---- online softmax (per query column == per lane) ---------------------------------
        float tmax = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kt * KT + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kv >= Nk) s[jb][r] = -INFINITY;
                tmax = fmaxf(tmax, s[jb][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);
        const float mb = m_new * scale_log2;
        float psum = 0.f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[jb][r], scale_log2, -mb));
                s[jb][r] = pv;
                psum += pv;
            }
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

        // ---- O^T += V^T P^T ---------------------------------------------------------------
        const char* vt = lds + C::TILE;
        if constexpr (sizeof(T) == 2) {
            // transposed read: per 16-lane group a 4x16 block; lane 4q+p supplies row q, cols 4p..4p+3
            const int i16 = lane & 15, g = lane >> 4;
            const int trow = 4 * (g >> 1) + (i16 >> 2);            // 4*half + q'
            const int tcol = 16 * (g & 1) + 4 * (i16 & 3);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (bf16)s[jb][8 * s2 + j];
                    const char* vbase = vt + (jb * 32 + 16 * s2 + trow) * C::RS + tcol * 2;
#pragma unroll
                    for (int db = 0; db < C::NDB; ++db) {
                        if (db * 32 < D) {
                            const char* pa = vbase + db * 64;
                            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4*)(pa));
                            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4*)(pa + 8 * C::RS));
                            bf16x8 vf;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const char* vrow = vt + row * C::RS + l31 * 4;
#pragma unroll
                    for (int db = 0; db < C::NDB; ++db) {
                        if (db * 32 < D) {
                            const float a = *reinterpret_cast<const float*>(vrow + db * 128);
                            o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[jb][r], o[db], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= inv;
}

// grid (ceil(Nq/128), H, B)
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs p, const float scale_log2) {
    typedef ACfg<T, DP> C;
    typedef typename FragOf<T>::type Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 128 + wave * 32 + l31;
    const int qc = q < p.Nq ? q : p.Nq - 1;
    const T* qrow = (const T*)p.q + ((size_t)b * p.Nq + qc) * p.ldq + h * p.D;
    QFrags<T, DP> qf;
    load_q<T, DP>(qf, qrow, p.D, half);
    const size_t kvoff = (size_t)(b % p.Bkv) * p.Nk * p.ldk + h * p.D;
    OAcc<T, DP> oa;
    attend<T, DP>(qf, (const T*)p.k + kvoff, (const T*)p.v + kvoff, p.ldk, p.Nk, p.D, scale_log2, smem, oa);
    auto& o = oa.b;
    if (q < p.Nq) {
        T* orow = (T*)p.out + ((size_t)b * p.Nq + q) * p.ldo + h * p.D;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * half;
                if (d < p.D) {
                    if constexpr (sizeof(T) == 2) {
                        bf16x4 v4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = (bf16)o[db][4 * g + j];
                        *reinterpret_cast<bf16x4*>(orow + d) = v4;
                    } else {
                        f32x4 v4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = o[db][4 * g + j];
                        *reinterpret_cast<f32x4*>(orow + d) = v4;
                    }
                }
            }
    }
}

// ---- fused score tail ----------------------------------------------------------------------
// grid (ceil(N/128), B*H, n_pairs*2); partial layout [pair][dir][bh][qtile][4] f32
template <typename T, int DP>
__global__ __launch_bounds__(256) void pair_tail_kernel(const T* __restrict__ qg, const T* __restrict__ kg,
                                                        const T* __restrict__ vg, const int32_t* __restrict__ idx_a,
                                                        const int32_t* __restrict__ idx_b, int B, int H, int N, int D,
                                                        float scale_log2, int mse, float* __restrict__ part) {
    typedef ACfg<T, DP> C;
    typedef typename FragOf<T>::type Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int pair = blockIdx.z >> 1, dir = blockIdx.z & 1;
    const int ia = idx_a[pair], ib = idx_b[pair];
    const int iq = dir ? ib : ia;        // query image (also the "self" keys/values)
    const int ix = dir ? ia : ib;        // the other image ("cross" keys/values)
    const int ld = H * D;
    const size_t img = (size_t)B * N * ld;
    const int q = blockIdx.x * 128 + wave * 32 + l31;
    const int qc = q < N ? q : N - 1;
    const size_t boff = (size_t)b * N * ld + h * D;
    QFrags<T, DP> qf;
    load_q<T, DP>(qf, qg + iq * img + boff + (size_t)qc * ld, D, half);
    OAcc<T, DP> osa, oxa;
    attend<T, DP>(qf, kg + iq * img + boff, vg + iq * img + boff, ld, N, D, scale_log2, smem, osa);
    attend<T, DP>(qf, kg + ix * img + boff, vg + ix * img + boff, ld, N, D, scale_log2, smem, oxa);
    auto& os = osa.b;
    auto& ox = oxa.b;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (q < N) {
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (d < D) {
                    const float x = ox[db][r], y = os[db][r];
                    if (mse) { const float df = x - y; s0 = fmaf(df, df, s0); }
                    else { s0 = fmaf(x, y, s0); s1 = fmaf(x, x, s1); s2 = fmaf(y, y, s2); }
                }
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + ((((size_t)pair * 2 + dir) * gridDim.y + bh) * gridDim.x + blockIdx.x) * 4;
        o[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        o[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        o[2] = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
        o[3] = 0.f;
    }
}

// one thread per pair: fixed-order f64 fold of the partials, then cosine / mse and the mean of
// the two directions (diffsim.py:187-197; F.cosine_similarity eps = 1e-8)
__global__ void pair_finish_kernel(const float* __restrict__ part, int n_pairs, int nblk, int mse, double count,
                                   float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    double res = 0.0;
    for (int dir = 0; dir < 2; ++dir) {
        double a = 0.0, x2 = 0.0, y2 = 0.0;
        const float* o = part + ((size_t)p * 2 + dir) * nblk * 4;
        for (int i = 0; i < nblk; ++i) { a += o[4 * i]; x2 += o[4 * i + 1]; y2 += o[4 * i + 2]; }
        if (mse) res += a / count;
        else {
            const double nx = sqrt(x2), ny = sqrt(y2);
            res += a / (fmax(nx, 1e-8) * fmax(ny, 1e-8));
        }
    }
    out[p] = (float)(res * 0.5);
}

template <typename T, int DP>
int launch_attn_dp(const AttnArgs& a, hipStream_t s) {
    typedef ACfg<T, DP> C;
    static bool attr_done = false;
    auto kern = attn_kernel<T, DP>;
    if (!attr_done) {
        DSIM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr_done = true;
    }
    const float scale_log2 = (1.0f / sqrtf((float)a.D)) * 1.4426950408889634f;
    hipLaunchKernelGGL(kern, dim3((a.Nq + 127) / 128, a.H, a.B), dim3(256), C::LDS, s, a, scale_log2);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

template <typename T>
int launch_attn_t(const AttnArgs& a, hipStream_t s) {
    if (a.D <= 32) return launch_attn_dp<T, 32>(a, s);
    if (a.D <= 64) return launch_attn_dp<T, 64>(a, s);
    if (a.D <= 96) return launch_attn_dp<T, 96>(a, s);
    if (a.D <= 160) return launch_attn_dp<T, 160>(a, s);
    return DSIM_ERR_INVALID;
}

template <typename T, int DP>
int launch_tail_dp(const void* q, const void* k, const void* v, const int32_t* ia, const int32_t* ib, int n_pairs,
                   int B, int H, int N, int D, int mse, float* out, void* scratch, hipStream_t s) {
    typedef ACfg<T, DP> C;
    static bool attr_done = false;
    auto kern = pair_tail_kernel<T, DP>;
    if (!attr_done) {
        DSIM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr_done = true;
    }
    const float scale_log2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
    const int qt = (N + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(qt, B * H, n_pairs * 2), dim3(256), C::LDS, s, (const T*)q, (const T*)k,
                       (const T*)v, ia, ib, B, H, N, D, scale_log2, mse, (float*)scratch);
    hipLaunchKernelGGL(pair_finish_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, s, (const float*)scratch, n_pairs,
                       qt * B * H, mse, (double)B * H * N * D, out);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

template <typename T>
int launch_tail_t(const void* q, const void* k, const void* v, const int32_t* ia, const int32_t* ib, int n_pairs,
                  int B, int H, int N, int D, int mse, float* out, void* scratch, hipStream_t s) {
    if (D <= 32) return launch_tail_dp<T, 32>(q, k, v, ia, ib, n_pairs, B, H, N, D, mse, out, scratch, s);
    if (D <= 64) return launch_tail_dp<T, 64>(q, k, v, ia, ib, n_pairs, B, H, N, D, mse, out, scratch, s);
    if (D <= 96) return launch_tail_dp<T, 96>(q, k, v, ia, ib, n_pairs, B, H, N, D, mse, out, scratch, s);
    if (D <= 160) return launch_tail_dp<T, 160>(q, k, v, ia, ib, n_pairs, B, H, N, D, mse, out, scratch, s);
    return DSIM_ERR_INVALID;
}

}  // namespace

int launch_attention(const AttnArgs& a, int dtype, hipStream_t s) {
    const int vec = dtype == DSIM_F32 ? 4 : 8;
    if (a.D % 8 || a.ldq % vec || a.ldk % vec || a.ldo % 4 || a.Nk < 1 || a.Nq < 1 || a.Bkv < 1)
        return DSIM_ERR_INVALID;
    if (dtype == DSIM_BF16) return launch_attn_t<bf16>(a, s);
    if (dtype == DSIM_F32) return launch_attn_t<float>(a, s);
    return DSIM_ERR_INVALID;
}

size_t pair_score_scratch_bytes(int n_pairs, int B, int H, int N, int D) {
    (void)D;
    return (size_t)n_pairs * 2 * B * H * ((N + 127) / 128) * 4 * sizeof(float);
}

int launch_pair_score(const void* q, const void* k, const void* v, const int32_t* ia, const int32_t* ib,
                      int n_pairs, int B, int H, int N, int D, int dtype, int similarity, float* out, void* scratch,
                      size_t scratch_bytes, hipStream_t s) {
    if (n_pairs <= 0 || D % 8 || N < 1) return DSIM_ERR_INVALID;
    if (scratch_bytes < pair_score_scratch_bytes(n_pairs, B, H, N, D)) return DSIM_ERR_WORKSPACE;
    if (n_pairs * 2 > 65535) return DSIM_ERR_INVALID;
    if (dtype == DSIM_BF16) return launch_tail_t<bf16>(q, k, v, ia, ib, n_pairs, B, H, N, D, similarity, out, scratch, s);
    if (dtype == DSIM_F32) return launch_tail_t<float>(q, k, v, ia, ib, n_pairs, B, H, N, D, similarity, out, scratch, s);
    return DSIM_ERR_INVALID;
}

}  // namespace dsim
