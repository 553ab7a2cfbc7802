// GroupNorm(+SiLU) and LayerNorm for token-major activations on gfx950 (HBM-bound kernels).
//
// Replaces torch.nn.GroupNorm / F.silu / torch.nn.LayerNorm as used by diffusers'
// ResnetBlock2D, Transformer2DModel and BasicTransformerBlock (SURVEY.md Appendix A items 3-5;
// block control flow: /root/reference/diffsim/hacked_modules.py:39-40, 336-339).
//
// GroupNorm runs as two launches over [B][HW][C] (C = C0+C1: the up-path channel concat
// cat([h, skip]) is read from its two sources and never materialised un-normalised):
//   1. gn_stats : every workgroup streams a slab of rows with 16-byte loads, each thread owning
//                 fixed channel slots; per-channel f32 partials -> fixed-order f64 reduction
//                 per group -> one (sum, sumsq) f64 pair per (batch, slab, group).
//   2. gn_apply : folds the slab partials in fixed order, then y = (x-mean)*rstd*gamma+beta
//                 [*sigmoid] with 16-byte loads and stores.
// No float atomics anywhere: results are bit-reproducible and independent of batch size.
#include <cstdlib>

#include "common.h"

namespace dsim {
// development A/B: DSIM_GN_ONEPASS=0 keeps the two-pass kernels at every level
#ifdef DSIM_DEVTOOLS
int g_gn_onepass = [] { const char* e = getenv("DSIM_GN_ONEPASS"); return e ? atoi(e) : 1; }();
int g_ln_rows = [] { const char* e = getenv("DSIM_LN_ROWS"); return e ? atoi(e) : 1; }();
#endif
namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SLOTS = 4;     // channel slots (16 B each) a thread may own: C <= 4*256*VEC

// x*sigmoid(x) with the hardware exp2/rcp (1 ulp each): the apply pass is otherwise VALU-bound on libm's expf + divide
__device__ __forceinline__ float silu_fast(float y) {
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
}

template <typename T> struct Vec16;
template <> struct Vec16<h16> { typedef h16x8 type; static constexpr int N = 8; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

__host__ __device__ inline int gn_chunks(int HW) {
    int c = HW / 64;
    c = c < 1 ? 1 : (c > 32 ? 32 : c);
    if (HW > 16384) c = 64;          // the VAE's 256^2 / 512^2 maps: few images, so more slabs per image
    return c;
}

template <typename T>
__device__ __forceinline__ typename Vec16<T>::type load_slot(const T* x0, int C0, const T* x1, int C1,
                                                             size_t row, int ch) {
    typedef typename Vec16<T>::type V;
    return ch < C0 ? *reinterpret_cast<const V*>(x0 + row * C0 + ch)
                   : *reinterpret_cast<const V*>(x1 + row * C1 + (ch - C0));
}

// grid (chunks, B); NS = channel slots per thread, UNR = rows in flight per thread
template <typename T, int NS, int UNR>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* __restrict__ x0, int C0,
                                                              const T* __restrict__ x1, int C1, int HW,
                                                              int groups, double* __restrict__ part) {
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = C0 + C1, S = C / VEC;
    const int tpr = S < GN_THREADS ? S : GN_THREADS;          // threads per row
    const int R = GN_THREADS / tpr;                            // rows in flight
    const int tid = threadIdx.x;
    const int trow = tid / tpr, tcol = tid - trow * tpr;
    const int chunks = gridDim.x, chunk = blockIdx.x, b = blockIdx.y;
    const int r0 = (int)((long)HW * chunk / chunks), r1 = (int)((long)HW * (chunk + 1) / chunks);

    float s1[NS][VEC], s2[NS][VEC];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s1[k][e] = s2[k][e] = 0.f;

    if (trow < R) {
        // UNR rows in flight per thread: all loads of a batch are issued before the first is consumed
        typedef typename Vec16<T>::type V;
        int r = r0 + trow;
        for (; r + (UNR - 1) * R < r1; r += UNR * R) {
            V v[UNR][NS];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int slot = tcol + k * tpr;
                    if (slot < S) v[u][k] = load_slot<T>(x0, C0, x1, C1, (size_t)b * HW + r + u * R, slot * VEC);
                }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int slot = tcol + k * tpr;
                    if (slot < S) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float f = (float)v[u][k][e];
                            s1[k][e] += f;
                            s2[k][e] = fmaf(f, f, s2[k][e]);
                        }
                    }
                }
        }
        for (; r < r1; r += R) {
            const size_t row = (size_t)b * HW + r;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int slot = tcol + k * tpr;
                if (slot < S) {
                    auto v = load_slot<T>(x0, C0, x1, C1, row, slot * VEC);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float f = (float)v[e];
                        s1[k][e] += f;
                        s2[k][e] = fmaf(f, f, s2[k][e]);
                    }
                }
            }
        }
    }
    // per-channel partials of the R row lanes -> LDS [R][C][2]
    float* lds = reinterpret_cast<float*>(smem);
    if (trow < R) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int slot = tcol + k * tpr;
            if (slot < S) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    lds[((size_t)trow * C + slot * VEC + e) * 2 + 0] = s1[k][e];
                    lds[((size_t)trow * C + slot * VEC + e) * 2 + 1] = s2[k][e];
                }
            }
        }
    }
    __syncthreads();
    // GN_THREADS / gp2 threads per group (gp2 = groups rounded up to a power of two) fold a fixed strided subset of the
    // group's R x cpg partials in f64, then a fixed xor-shuffle tree: same order for a given shape, and no single thread
    // walking the whole list while the workgroup waits
    int gp2 = 1;
    while (gp2 < groups) gp2 *= 2;
    const int tg = GN_THREADS / gp2;                           // >= 4 (groups <= 64)
    {
        const int cpg = C / groups;
        const int g = tid / tg, t = tid - g * tg, cnt = R * cpg;
        double a = 0.0, q = 0.0;
        if (g < groups)
            for (int i = t; i < cnt; i += tg) {
                const int rr = i / cpg, c = g * cpg + (i - rr * cpg);
                const float2 pr = *reinterpret_cast<const float2*>(&lds[((size_t)rr * C + c) * 2]);
                a += (double)pr.x;
                q += (double)pr.y;
            }
        for (int off = tg >> 1; off > 0; off >>= 1) {          // tg <= 64 whenever groups >= 4; wider: see below
            if (off < 64) {
                a += __shfl_xor(a, off, 64);
                q += __shfl_xor(q, off, 64);
            }
        }
        if (tg > 64) {                                         // fewer than 4 groups: a group spans tg / 64 waves
            __shared__ double s_w[4][2];
            if ((tid & 63) == 0) { s_w[tid >> 6][0] = a; s_w[tid >> 6][1] = q; }
            __syncthreads();
            if (t == 0) {
                a = 0.0; q = 0.0;
                for (int k = 0; k < tg / 64; ++k) { a += s_w[g * (tg / 64) + k][0]; q += s_w[g * (tg / 64) + k][1]; }
            }
        }
        if (t == 0 && g < groups) {
            double* o = part + (((size_t)b * chunks + chunk) * groups + g) * 2;
            o[0] = a;
            o[1] = q;
        }
    }
}

// Statistics from the producing conv's epilogue (gemm.hip, GemmArgs.gn_part): part32[b][chunk][quad][2] f32 = (sum, sum of squares) of
// one wave's 64 rows x one 4-channel quad, chunk = (256-row tile of the image) x 4 + wave row.  One workgroup per (group, image) folds
// them into out[b][0][g][2] (f64; the layout gn_apply_kernel reads with chunks = 1): thread t takes chunks t, t + 256, ... (the
// group's quads in order inside a chunk), then a fixed xor-shuffle tree per wave and the four waves in order.
__global__ __launch_bounds__(GN_THREADS) void gn_fold_kernel(const float* __restrict__ part32, int chunks, int quads, int groups,
                                                             double* __restrict__ out) {
    __shared__ double s_w[GN_THREADS / 64][2];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int qpg = quads / groups;
    double a = 0.0, q = 0.0;
    const float* src = part32 + ((size_t)b * chunks * quads + (size_t)g * qpg) * 2;
    for (int c = tid; c < chunks; c += GN_THREADS)
        for (int k = 0; k < qpg; ++k) {
            const float2 pr = *reinterpret_cast<const float2*>(src + ((size_t)c * quads + k) * 2);
            a += (double)pr.x;
            q += (double)pr.y;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        q += __shfl_xor(q, off, 64);
    }
    if ((tid & 63) == 0) { s_w[tid >> 6][0] = a; s_w[tid >> 6][1] = q; }
    __syncthreads();
    if (tid == 0) {
        a = 0.0; q = 0.0;
        for (int w = 0; w < GN_THREADS / 64; ++w) { a += s_w[w][0]; q += s_w[w][1]; }
        out[((size_t)b * groups + g) * 2] = a;
        out[((size_t)b * groups + g) * 2 + 1] = q;
    }
}

// grid (row_blocks, B)
template <typename T, bool SILU, int NS, int UNR>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* __restrict__ x0, int C0,
                                                              const T* __restrict__ x1, int C1,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              T* __restrict__ out, int HW, int groups,
                                                              float eps, int chunks,
                                                              const double* __restrict__ part) {
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::type V;
    __shared__ float s_mean[64], s_rstd[64];
    const int C = C0 + C1, S = C / VEC;
    const int tpr = S < GN_THREADS ? S : GN_THREADS;
    const int R = GN_THREADS / tpr;
    const int tid = threadIdx.x;
    const int trow = tid / tpr, tcol = tid - trow * tpr;
    const int b = blockIdx.y;
    const int cpg = C / groups;
    // the image's chunks x groups partial pairs: fetched by the whole workgroup in one round trip (a per-group
    // thread walking them serially cost ~32 dependent L2 round trips before the first row was streamed), then folded
    // by one thread per group in fixed order
    extern __shared__ __attribute__((aligned(16))) char smem_apply[];
    double* fold = reinterpret_cast<double*>(smem_apply);
    {
        const double* src = part + (size_t)b * chunks * groups * 2;
        const int cnt = chunks * groups * 2;
        for (int i = tid; i < cnt; i += GN_THREADS) fold[i] = src[i];
    }
    __syncthreads();
    {
        // 4 threads per group (groups <= 64) take every 4th chunk, then a two-step xor-shuffle: fixed order per shape
        const int g = tid >> 2, t = tid & 3;
        double a = 0.0, q = 0.0;
        if (g < groups)
            for (int c = t; c < chunks; c += 4) {
                a += fold[(c * groups + g) * 2];
                q += fold[(c * groups + g) * 2 + 1];
            }
        a += __shfl_xor(a, 2, 64); q += __shfl_xor(q, 2, 64);
        a += __shfl_xor(a, 1, 64); q += __shfl_xor(q, 1, 64);
        if (t == 0 && g < groups) {
            const double n = (double)HW * cpg;
            const double mean = a / n;
            double var = q / n - mean * mean;
            if (var < 0.0) var = 0.0;
            s_mean[g] = (float)mean;
            s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    if (trow >= R) return;
    float sc[NS][VEC], sh[NS][VEC];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int slot = tcol + k * tpr;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (slot < S) {
                const int c = slot * VEC + e, g = c / cpg;
                const float w = gamma[c] * s_rstd[g];
                sc[k][e] = w;
                sh[k][e] = beta[c] - s_mean[g] * w;
            } else {
                sc[k][e] = sh[k][e] = 0.f;
            }
        }
    }
    const int rows_per_block = (HW + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = (r0 + rows_per_block < HW) ? r0 + rows_per_block : HW;
    int r = r0 + trow;
    for (; r + (UNR - 1) * R < r1; r += UNR * R) {
        V v[UNR][NS];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int slot = tcol + k * tpr;
                if (slot < S) v[u][k] = load_slot<T>(x0, C0, x1, C1, (size_t)b * HW + r + u * R, slot * VEC);
            }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int slot = tcol + k * tpr;
                if (slot < S) {
                    V o;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        float y = fmaf((float)v[u][k][e], sc[k][e], sh[k][e]);
                        if (SILU) y = silu_fast(y);
                        o[e] = (T)y;
                    }
                    *reinterpret_cast<V*>(out + ((size_t)b * HW + r + u * R) * C + slot * VEC) = o;
                }
            }
    }
    for (; r < r1; r += R) {
        const size_t row = (size_t)b * HW + r;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int slot = tcol + k * tpr;
            if (slot < S) {
                V v = load_slot<T>(x0, C0, x1, C1, row, slot * VEC);
                V o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float y = fmaf((float)v[e], sc[k][e], sh[k][e]);
                    if (SILU) y = silu_fast(y);
                    o[e] = (T)y;
                }
                *reinterpret_cast<V*>(out + row * C + slot * VEC) = o;
            }
        }
    }
}

// one wave per RPW consecutive rows (all RPW*MAXS loads of a wave are issued before any reduction, so narrow rows
// -- C = 320 fills only 40 of 64 lanes -- still keep enough bytes in flight); 4 waves per workgroup.
// MOD: no affine; y = xhat * (1 + scale[half][c]) + shift[half][c] with gamma = scale, beta = shift, [2][C] each
template <typename T, bool MOD, int MAXS, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out,
                                                        int M, int C, float eps, int rows_per_batch) {
    constexpr int VEC = Vec16<T>::N;        // C <= 64*MAXS*VEC
    typedef typename Vec16<T>::type V;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= M) return;
    const int S = C / VEC;
    V t[RPW][MAXS];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = row0 + rr < M ? row0 + rr : M - 1;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int slot = lane + k * 64;
            if (slot < S) t[rr][k] = *reinterpret_cast<const V*>(x + (size_t)row * C + slot * VEC);
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = row0 + rr;
        if (row >= M) break;
        float v[MAXS][VEC];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int slot = lane + k * 64;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v[k][e] = slot < S ? (float)t[rr][k][e] : 0.f;
                sum += v[k][e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int slot = lane + k * 64;
            if (slot < S) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float d = v[k][e] - mean; sq = fmaf(d, d, sq); }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
        T* orow = out + (size_t)row * C;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int slot = lane + k * 64;
            if (slot < S) {
                V o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int c = slot * VEC + e;
                    if (MOD) {
                        const int hoff = ((row / rows_per_batch) & 1) * C;
                        o[e] = (T)fmaf((v[k][e] - mean) * rstd, 1.0f + gamma[hoff + c], beta[hoff + c]);
                    } else {
                        o[e] = (T)fmaf((v[k][e] - mean) * rstd, gamma[c], beta[c]);
                    }
                }
                *reinterpret_cast<V*>(orow + slot * VEC) = o;
            }
        }
    }
}

// Affine LayerNorm with LPR lanes per row (LPR = the largest power of two dividing the row's S 16-byte chunks, CPL = S / LPR
// chunks per lane): a wave pass covers 64 / LPR rows with all 64 lanes busy, lane s of a row owning chunks s, s + LPR, ... so
// every load instruction reads runs of LPR * 16 contiguous bytes.  C = 320 (40 chunks) keeps 8 lanes x 5 chunks per row
// instead of 40 of 64 lanes and a six-step cross-lane reduction per row: 2.4x fewer VALU instructions per byte, which is
// what bounded the wave-per-row form at the 64 x 64 level.  gamma / beta sit in LDS; the next pass's rows are in flight
// while the current pass is reduced.  Two-pass statistics (mean, then centred squares), fixed order per shape.
template <typename T, int CPL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, T* __restrict__ out,
                                                             int M, int C, float eps, int LPR, int passes) {
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::type V;
    extern __shared__ __attribute__((aligned(16))) char smem_ln[];
    float* s_g = reinterpret_cast<float*>(smem_ln);
    float* s_b = s_g + C;
    for (int i = threadIdx.x; i < C; i += 256) { s_g[i] = gamma[i]; s_b[i] = beta[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / LPR;                                  // rows per wave pass
    const int sub = lane & (LPR - 1), rl = lane / LPR;
    const int base = (blockIdx.x * 4 + wave) * passes * rpw;
    if (base >= M) return;
    const float invC = 1.0f / (float)C;
    V t[CPL], tn[CPL];
    {
        const int row = base + rl < M ? base + rl : M - 1;
#pragma unroll
        for (int k = 0; k < CPL; ++k) t[k] = *reinterpret_cast<const V*>(x + (size_t)row * C + (sub + LPR * k) * VEC);
    }
    for (int p = 0; p < passes; ++p) {
        const int row = base + p * rpw + rl;
        if (base + p * rpw >= M) break;                        // wave-uniform
        if (p + 1 < passes) {
            const int nr = row + rpw < M ? row + rpw : M - 1;
#pragma unroll
            for (int k = 0; k < CPL; ++k) tn[k] = *reinterpret_cast<const V*>(x + (size_t)nr * C + (sub + LPR * k) * VEC);
        }
        float v[CPL][VEC];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { v[k][e] = (float)t[k][e]; sum += v[k][e]; }
        if (LPR > 32) sum += __shfl_xor(sum, 32);
        if (LPR > 16) sum += __shfl_xor(sum, 16);
        if (LPR > 8) sum += __shfl_xor(sum, 8);
        if (LPR > 4) sum += __shfl_xor(sum, 4);
        if (LPR > 2) sum += __shfl_xor(sum, 2);
        if (LPR > 1) sum += __shfl_xor(sum, 1);
        const float mean = sum * invC;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { v[k][e] -= mean; sq = fmaf(v[k][e], v[k][e], sq); }
        if (LPR > 32) sq += __shfl_xor(sq, 32);
        if (LPR > 16) sq += __shfl_xor(sq, 16);
        if (LPR > 8) sq += __shfl_xor(sq, 8);
        if (LPR > 4) sq += __shfl_xor(sq, 4);
        if (LPR > 2) sq += __shfl_xor(sq, 2);
        if (LPR > 1) sq += __shfl_xor(sq, 1);
        const float rstd = 1.0f / sqrtf(sq * invC + eps);
        if (row < M) {
            T* orow = out + (size_t)row * C;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = (sub + LPR * k) * VEC;
                V o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = (T)fmaf(v[k][e] * rstd, s_g[c + e], s_b[c + e]);
                *reinterpret_cast<V*>(orow + c) = o;
            }
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) t[k] = tn[k];
    }
}

template <typename T, int CPL>
void ln_rows_launch(const void* x, const float* gamma, const float* beta, void* out, int M, int C, float eps, int LPR,
                    hipStream_t s) {
    const int rpw = 64 / LPR;
    int passes = 4;                                             // rows per workgroup = 4 waves x passes x rpw
    while (passes > 1 && (M + 4 * passes * rpw - 1) / (4 * passes * rpw) < 2048) passes >>= 1;
    const int blocks = (M + 4 * passes * rpw - 1) / (4 * passes * rpw);
    hipLaunchKernelGGL((layernorm_rows_kernel<T, CPL>), dim3(blocks), dim3(256), (size_t)2 * C * sizeof(float), s,
                       (const T*)x, gamma, beta, (T*)out, M, C, eps, LPR, passes);
}

template <typename T, bool MOD>
int ln_typed(const void* x, const float* gamma, const float* beta, void* out, int M, int C, float eps, int rpb,
             hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    if (C % VEC || C > 64 * 6 * VEC || M < 1) return DSIM_ERR_INVALID;
    const int S = C / VEC;
    const dim3 block(256);
    if (!MOD && g_ln_rows && S <= 80) {      // wider rows: the wave-per-row form below already streams at > 6 TB/s
        int LPR = 1;
        while (LPR < 64 && S % (LPR * 2) == 0) LPR *= 2;
        const int CPL = S / LPR;                       // odd by construction
        bool done = true;
        switch (CPL) {
            case 1: ln_rows_launch<T, 1>(x, gamma, beta, out, M, C, eps, LPR, s); break;
            case 3: ln_rows_launch<T, 3>(x, gamma, beta, out, M, C, eps, LPR, s); break;
            case 5: ln_rows_launch<T, 5>(x, gamma, beta, out, M, C, eps, LPR, s); break;
            default: done = false;
        }
        if (done) { DSIM_HIP_CHECK(hipGetLastError()); return DSIM_OK; }
    }
    if (S <= 64)
        hipLaunchKernelGGL((layernorm_kernel<T, MOD, 1, 8>), dim3((M + 31) / 32), block, 0, s, (const T*)x, gamma, beta,
                           (T*)out, M, C, eps, rpb);
    else if (S <= 128)
        hipLaunchKernelGGL((layernorm_kernel<T, MOD, 2, 2>), dim3((M + 7) / 8), block, 0, s, (const T*)x, gamma, beta,
                           (T*)out, M, C, eps, rpb);
    else if (S <= 192)
        hipLaunchKernelGGL((layernorm_kernel<T, MOD, 3, 2>), dim3((M + 7) / 8), block, 0, s, (const T*)x, gamma, beta,
                           (T*)out, M, C, eps, rpb);
    else
        hipLaunchKernelGGL((layernorm_kernel<T, MOD, 6, 1>), dim3((M + 3) / 4), block, 0, s, (const T*)x, gamma, beta,
                           (T*)out, M, C, eps, rpb);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

// softmax over rows of `cols` elements (one 256-thread workgroup per row; cols % VEC == 0).  Used by
// the VAE encoder's single-head 512-d mid-block attention, whose score matrix is materialised by
// the GEMM kernel.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ out, int cols,
                                                           float scale_log2) {
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::type V;
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* xr = x + (size_t)blockIdx.x * cols;
    T* orow = out + (size_t)blockIdx.x * cols;
    const int S = cols / VEC;
    float m = -INFINITY;
    for (int s = tid; s < S; s += 256) {
        const V v = *reinterpret_cast<const V*>(xr + s * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) m = fmaxf(m, (float)v[e]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mb = m * scale_log2;
    float sum = 0.f;
    for (int s = tid; s < S; s += 256) {
        const V v = *reinterpret_cast<const V*>(xr + s * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) sum += exp2f(fmaf((float)v[e], scale_log2, -mb));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int s = tid; s < S; s += 256) {
        const V v = *reinterpret_cast<const V*>(xr + s * VEC);
        V o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (T)(exp2f(fmaf((float)v[e], scale_log2, -mb)) * inv);
        *reinterpret_cast<V*>(orow + s * VEC) = o;
    }
}

// One-pass GroupNorm for the low-resolution levels (16 x 16 and 8 x 8 maps: 52 of the 83 GroupNorm launches of an SD1.5
// forward, 35 % of their time): a workgroup takes ALL HW rows of a channel slab made of whole groups of one image and keeps
// them in registers (<= MAXCH 16-byte chunks per thread), so the tensor is read once instead of twice and one launch
// replaces two.  grid (P slabs, B); a thread owns one chunk column and every R-th row, exactly like the two-pass kernels,
// and the statistics use the same per-channel f32 partials -> fixed-order f64 fold per group (bit-reproducible,
// independent of the batch size).
template <typename T, bool SILU, int MAXCH>
__global__ __launch_bounds__(GN_THREADS) void gn_onepass_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1,
                                                                int C1, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, T* __restrict__ out, int HW,
                                                                int groups, float eps, int CS) {
    constexpr int VEC = Vec16<T>::N;
    typedef typename Vec16<T>::type V;
    extern __shared__ __attribute__((aligned(16))) char smem_op[];
    __shared__ float s_mean[64], s_rstd[64];
    const int C = C0 + C1, cpg = C / groups;
    const int tpr = CS / VEC;                                  // threads per row (chunk columns of the slab)
    const int R = GN_THREADS / tpr;                            // rows in flight
    const int tid = threadIdx.x, trow = tid / tpr, tcol = tid - trow * tpr;
    const int b = blockIdx.y, c0 = blockIdx.x * CS;            // first channel of this slab
    const int ch = c0 + tcol * VEC;                            // first channel of this thread's chunks
    const bool live = trow < R;
    V v[MAXCH];
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
    if (live) {
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int r = trow + i * R;
            if (r < HW) v[i] = load_slot<T>(x0, C0, x1, C1, (size_t)b * HW + r, ch);
        }
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int r = trow + i * R;
            if (r < HW) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float f = (float)v[i][e];
                    s1[e] += f;
                    s2[e] = fmaf(f, f, s2[e]);
                }
            }
        }
    }
    float* lds = reinterpret_cast<float*>(smem_op);            // [R][CS][2]
    if (live) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            lds[((size_t)trow * CS + tcol * VEC + e) * 2 + 0] = s1[e];
            lds[((size_t)trow * CS + tcol * VEC + e) * 2 + 1] = s2[e];
        }
    }
    __syncthreads();
    // fold of a group's R x cpg channel partials in f64: tg = GN_THREADS / gslab threads per group take a fixed strided
    // subset each, then a fixed xor-shuffle tree (and, for groups wider than a wave, a last serial fold of the per-wave
    // sums).  The order depends on the shape only.  (One thread per group walking the list serially held the whole
    // workgroup -- and, with every workgroup of the launch in the same phase, the chip -- for several microseconds.)
    const int gslab = CS / cpg;                                // whole groups in this slab (a power of two)
    const int tg = GN_THREADS / gslab;                         // threads per group: 8 .. 256
    {
        const int g = tid / tg, t = tid - g * tg, cnt = R * cpg;
        double a = 0.0, q = 0.0;
        for (int i = t; i < cnt; i += tg) {
            const int rr = i / cpg, c = g * cpg + (i - rr * cpg);
            const float2 pr = *reinterpret_cast<const float2*>(&lds[((size_t)rr * CS + c) * 2]);
            a += (double)pr.x;
            q += (double)pr.y;
        }
        const int w = tg < 64 ? tg : 64;
        for (int off = w >> 1; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            q += __shfl_xor(q, off, 64);
        }
        __shared__ double s_wsum[4][2];                        // per-wave sums when a group spans several waves
        if (tg > 64) {
            if ((tid & 63) == 0) { s_wsum[tid >> 6][0] = a; s_wsum[tid >> 6][1] = q; }
            __syncthreads();
            if (t == 0) {
                a = 0.0; q = 0.0;
                for (int k = 0; k < tg / 64; ++k) { a += s_wsum[g * (tg / 64) + k][0]; q += s_wsum[g * (tg / 64) + k][1]; }
            }
        }
        if (t == 0) {
            const double n = (double)HW * cpg;
            const double mean = a / n;
            double var = q / n - mean * mean;
            if (var < 0.0) var = 0.0;
            s_mean[g] = (float)mean;
            s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    if (!live) return;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int c = ch + e, g = (c - c0) / cpg;
        const float w = gamma[c] * s_rstd[g];
        sc[e] = w;
        sh[e] = beta[c] - s_mean[g] * w;
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int r = trow + i * R;
        if (r < HW) {
            V o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float y = fmaf((float)v[i][e], sc[e], sh[e]);
                if (SILU) y = silu_fast(y);
                o[e] = (T)y;
            }
            *reinterpret_cast<V*>(out + ((size_t)b * HW + r) * C + ch) = o;
        }
    }
}

// slab width (channels) of the one-pass form for this shape, or 0 when it does not apply: whole groups, 16-byte chunk
// columns of at least 256 B per row, no slab straddling the two concat sources, at most MAXCH chunks per thread
constexpr int GN_OP_MAXCH = 24;
template <typename T>
int gn_onepass_slab(int C0, int C1, int B, int HW, int groups) {
    constexpr int VEC = Vec16<T>::N;
    const int C = C0 + C1, cpg = C / groups;
    // The choice depends on the SHAPE only, never on the batch: slab width sets the summation grouping, and a batch of N
    // must score bit for bit like N single images.  Widest legal slab = longest coalesced row segments (narrower slabs
    // with more, shorter workgroups measured slower at the 8 x 8 level: 17.7 vs 16.8 us, 29 vs 22 us at 2560 channels).
    (void)B;
    int best = 0;
    for (int gs = 1; gs <= groups; gs *= 2) {                  // groups per slab
        const int CS = gs * cpg;
        if (groups % gs || CS % VEC || CS / VEC > GN_THREADS) continue;
        if (CS * (int)sizeof(T) < 256) continue;               // row segments shorter than 256 B waste the memory pipe
        if (C1 && (C0 % CS)) continue;                         // a slab may not straddle the concat boundary
        if (C / CS < 2) continue;                              // at least two workgroups per image
        const int tpr = CS / VEC, R = GN_THREADS / tpr;
        if ((HW + R - 1) / R > GN_OP_MAXCH) continue;
        if ((size_t)R * CS * 2 * sizeof(float) > 48 * 1024) continue;
        best = CS;
    }
    return best;
}

#ifdef DSIM_DEVTOOLS
// (g_norm_lds_pad: kbench occupancy probe, KB of unused LDS per GroupNorm workgroup; defined in attention.hip)
#define GN_PAD ((size_t)g_norm_lds_pad * 1024)
#else
#define GN_PAD ((size_t)0)
#endif

template <typename T, int NS, int UNR>
int gn_launch(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta, void* out, int B,
              int HW, int groups, float eps, int silu, void* scratch, int chunks, int rb, size_t lds, hipStream_t s,
              const float* pre = nullptr, int pre_chunks = 0) {
    if (pre) {
        // the statistics pass already happened in the producing conv's epilogue (GemmArgs.gn_part): fold its per-(wave, 4-channel quad)
        // f32 partials into one f64 pair per (image, group), in fixed order
        hipLaunchKernelGGL(gn_fold_kernel, dim3(groups, B), dim3(GN_THREADS), 0, s, pre, pre_chunks, (C0 + C1) / 4, groups, (double*)scratch);
        chunks = 1;
    } else
    hipLaunchKernelGGL((gn_stats_kernel<T, NS, UNR>), dim3(chunks, B), dim3(GN_THREADS), lds + GN_PAD, s, (const T*)x0, C0,
                       (const T*)x1, C1, HW, groups, (double*)scratch);
    const size_t alds = (size_t)chunks * groups * 2 * sizeof(double) + GN_PAD;       // <= 32 KB
    if (silu)
        hipLaunchKernelGGL((gn_apply_kernel<T, true, NS, UNR>), dim3(rb, B), dim3(GN_THREADS), alds, s, (const T*)x0, C0,
                           (const T*)x1, C1, gamma, beta, (T*)out, HW, groups, eps, chunks, (const double*)scratch);
    else
        hipLaunchKernelGGL((gn_apply_kernel<T, false, NS, UNR>), dim3(rb, B), dim3(GN_THREADS), alds, s, (const T*)x0, C0,
                           (const T*)x1, C1, gamma, beta, (T*)out, HW, groups, eps, chunks, (const double*)scratch);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

template <typename T>
int gn_typed(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta,
             void* out, int B, int HW, int groups, float eps, int silu, void* scratch, hipStream_t s,
             const float* pre = nullptr, int pre_chunks = 0) {
    constexpr int VEC = Vec16<T>::N;
    const int C = C0 + (x1 ? C1 : 0);
    if (!x1) C1 = 0;
    if (C % groups || C0 % VEC || C1 % VEC || groups > 64 || C > GN_MAX_SLOTS * GN_THREADS * VEC)
        return DSIM_ERR_INVALID;
    // (precomputed statistics: whole 4-channel quads per group)
    if (pre && (x1 || groups > 64 || (C / groups) % 4)) return DSIM_ERR_INVALID;
    if (const int CS = (g_gn_onepass && !pre) ? gn_onepass_slab<T>(C0, C1, B, HW, groups) : 0) {
        const int tpr1 = CS / VEC, R1 = GN_THREADS / tpr1;
        const size_t lds1 = (size_t)R1 * CS * 2 * sizeof(float);
        if (silu)
            hipLaunchKernelGGL((gn_onepass_kernel<T, true, GN_OP_MAXCH>), dim3(C / CS, B), dim3(GN_THREADS), lds1, s, (const T*)x0, C0,
                               (const T*)x1, C1, gamma, beta, (T*)out, HW, groups, eps, CS);
        else
            hipLaunchKernelGGL((gn_onepass_kernel<T, false, GN_OP_MAXCH>), dim3(C / CS, B), dim3(GN_THREADS), lds1, s, (const T*)x0, C0,
                               (const T*)x1, C1, gamma, beta, (T*)out, HW, groups, eps, CS);
        DSIM_HIP_CHECK(hipGetLastError());
        return DSIM_OK;
    }
    const int S = C / VEC, tpr = S < GN_THREADS ? S : GN_THREADS, R = GN_THREADS / tpr;
    const int chunks = gn_chunks(HW);
    const size_t lds = (size_t)R * C * 2 * sizeof(float);
    if (lds > 64 * 1024) return DSIM_ERR_INVALID;
    // row blocks per image of the apply pass (no effect on the numbers): about 1024 workgroups in all, so that large
    // batches amortise each workgroup's statistics fold over more rows and small batches still fill the chip
    int rb = HW / (R * 4);
    const int want = (1024 + B - 1) / B;
    if (rb > want) rb = want;
    rb = rb < 1 ? 1 : (rb > 64 ? 64 : rb);
    const int ns = (S + tpr - 1) / tpr;
    if (ns == 1)
        return gn_launch<T, 1, 4>(x0, C0, x1, C1, gamma, beta, out, B, HW, groups, eps, silu, scratch, chunks, rb, lds, s, pre, pre_chunks);
    if (ns == 2)
        return gn_launch<T, 2, 2>(x0, C0, x1, C1, gamma, beta, out, B, HW, groups, eps, silu, scratch, chunks, rb, lds, s, pre, pre_chunks);
    return gn_launch<T, GN_MAX_SLOTS, 1>(x0, C0, x1, C1, gamma, beta, out, B, HW, groups, eps, silu, scratch, chunks, rb,
                                         lds, s, pre, pre_chunks);
}

}  // namespace

// passes over the tensor the GroupNorm of this shape makes (2 = one-pass form: read + write; 3 = statistics read + read + write)
int groupnorm_passes(int C0, int C1, int HW, int groups, int dtype) {
    if (!g_gn_onepass) return 3;
    const int cs = dtype == DSIM_F32 ? gn_onepass_slab<float>(C0, C1, 1, HW, groups) : gn_onepass_slab<h16>(C0, C1, 1, HW, groups);   // (either 16-bit type)
    return cs ? 2 : 3;
}

size_t groupnorm_scratch_bytes(int B, int groups) { return (size_t)B * 64 * groups * 2 * sizeof(double); }

// GroupNorm(+SiLU) whose statistics pass already ran in the producing conv's epilogue: part32 = GemmArgs.gn_part of that launch,
// chunks = its partial rows per image (HW / 64); 16-bit dtypes only
int launch_groupnorm_pre(const void* x, int C, const float* gamma, const float* beta, void* out, int B, int HW, int groups, float eps,
                         int silu, int dtype, void* scratch, const float* part32, int chunks, hipStream_t s) {
    if (!part32 || chunks < 1) return DSIM_ERR_INVALID;
    if (dtype == DSIM_H16) return gn_typed<h16>(x, C, nullptr, 0, gamma, beta, out, B, HW, groups, eps, silu, scratch, s, part32, chunks);
#if !defined(DSIM_H16_IS_F16) && defined(DSIM_HAS_F16_TWINS)
    if (dtype == DSIM_F16) return launch_groupnorm_pre_f16(x, C, gamma, beta, out, B, HW, groups, eps, silu, dtype, scratch, part32, chunks, s);
#endif
    return DSIM_ERR_INVALID;
}

int launch_groupnorm(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta,
                     void* out, int B, int HW, int groups, float eps, int silu, int dtype, void* scratch,
                     hipStream_t s) {
    if (dtype == DSIM_H16)
        return gn_typed<h16>(x0, C0, x1, C1, gamma, beta, out, B, HW, groups, eps, silu, scratch, s);
#ifndef DSIM_H16_IS_F16
    if (dtype == DSIM_F32)
        return gn_typed<float>(x0, C0, x1, C1, gamma, beta, out, B, HW, groups, eps, silu, scratch, s);
#ifdef DSIM_HAS_F16_TWINS
    if (dtype == DSIM_F16)
        return launch_groupnorm_f16(x0, C0, x1, C1, gamma, beta, out, B, HW, groups, eps, silu, dtype, scratch, s);
#endif
#endif
    return DSIM_ERR_INVALID;
}

int launch_layernorm(const void* x, const float* gamma, const float* beta, void* out, int M, int C, float eps,
                     int dtype, hipStream_t s) {
    if (dtype == DSIM_H16) return ln_typed<h16, false>(x, gamma, beta, out, M, C, eps, 1, s);
#ifndef DSIM_H16_IS_F16
    if (dtype == DSIM_F32) return ln_typed<float, false>(x, gamma, beta, out, M, C, eps, 1, s);
#ifdef DSIM_HAS_F16_TWINS
    if (dtype == DSIM_F16) return launch_layernorm_f16(x, gamma, beta, out, M, C, eps, dtype, s);
#endif
#endif
    return DSIM_ERR_INVALID;
}

int launch_layernorm_mod(const void* x, const float* scale2, const float* shift2, void* out, int M, int C,
                         int rows_per_batch, float eps, int dtype, hipStream_t s) {
    if (rows_per_batch < 1) return DSIM_ERR_INVALID;
    if (dtype == DSIM_H16) return ln_typed<h16, true>(x, scale2, shift2, out, M, C, eps, rows_per_batch, s);
#ifndef DSIM_H16_IS_F16
    if (dtype == DSIM_F32) return ln_typed<float, true>(x, scale2, shift2, out, M, C, eps, rows_per_batch, s);
#ifdef DSIM_HAS_F16_TWINS
    if (dtype == DSIM_F16) return launch_layernorm_mod_f16(x, scale2, shift2, out, M, C, rows_per_batch, eps, dtype, s);
#endif
#endif
    return DSIM_ERR_INVALID;
}

int launch_softmax_rows(const void* x, void* out, int rows, int cols, float scale, int dtype, hipStream_t s) {
    const int vec = dtype == DSIM_F32 ? 4 : 8;
    if (cols % vec || rows < 1) return DSIM_ERR_INVALID;
    const float sl2 = scale * 1.4426950408889634f;
    if (dtype == DSIM_H16)
        hipLaunchKernelGGL(softmax_rows_kernel<h16>, dim3(rows), dim3(256), 0, s, (const h16*)x, (h16*)out, cols, sl2);
#ifndef DSIM_H16_IS_F16
    else if (dtype == DSIM_F32)
        hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)x, (float*)out, cols, sl2);
#ifdef DSIM_HAS_F16_TWINS
    else if (dtype == DSIM_F16)
        return launch_softmax_rows_f16(x, out, rows, cols, scale, dtype, s);
#endif
#endif
    else
        return DSIM_ERR_INVALID;
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

}  // namespace dsim
