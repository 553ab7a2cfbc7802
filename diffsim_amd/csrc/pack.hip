// Small utility kernels: weight repacking into the engine's layouts, the time-embedding GEMVs,
// noising + CFG duplication + conv_in, dtype conversion.  None of these is on the per-batch
// critical path except prep_conv_in (0.02 % of the FLOPs).
#include "common.h"

namespace dsim {
#ifdef DSIM_DEVTOOLS
int g_prep8 = [] { const char* e = getenv("DSIM_PREP8"); return e ? atoi(e) : 1; }();
#endif
namespace {

__device__ __forceinline__ float ld_any(const void* p, int dt, size_t i) {
    if (dt == DSIM_F32) return ((const float*)p)[i];
    if (dt == DSIM_BF16) return (float)((const bf16_t*)p)[i];
    return (float)((const _Float16*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int dt, size_t i, float v) {
    if (dt == DSIM_F32) ((float*)p)[i] = v;
    else if (dt == DSIM_BF16) ((bf16_t*)p)[i] = (bf16_t)v;
    else ((f16_t*)p)[i] = (f16_t)v;
}

// GEGLU row interleave: diffusers' ff.net.0.proj has rows [h (0..F) ; g (F..2F)].  The GEMM's
// GEGLU epilogue wants packed rows in alternating blocks of `rows` (32 or 16: geglu_block_rows()) [h blk0, g blk0, h blk1, ...].
__device__ __forceinline__ int geglu_src_row(int packed, int N, int rows) {
    const int F = N >> 1, blk = packed / rows, within = packed - blk * rows;
    const int j = (blk >> 1) * rows + within;
    return (blk & 1) ? F + j : j;
}

__global__ void pack_linear_kernel(const void* src, int sdt, void* dst, int ddt, int N, int K, int geglu) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (size_t)n * K);
    const int sn = geglu ? geglu_src_row(n, N, geglu) : n;
    st_any(dst, ddt, i, ld_any(src, sdt, (size_t)sn * K + k));
}

// [Cout][Cin][3][3] -> [Cout][tap = ky*3+kx][Cin]
__global__ void pack_conv3_kernel(const void* src, int sdt, void* dst, int ddt, int Cout, int Cin) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cout * 9 * Cin) return;
    const int ci = (int)(i % Cin);
    const int tap = (int)((i / Cin) % 9);
    const int co = (int)(i / ((size_t)9 * Cin));
    st_any(dst, ddt, i, ld_any(src, sdt, ((size_t)co * Cin + ci) * 9 + tap));
}

// conv_in for the direct kernel: [Cout][Cin][3][3] -> f32 [tap*Cin + ci][Cout]
__global__ void pack_conv_in_kernel(const void* src, int sdt, float* dst, int Cout, int Cin) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cout * 9 * Cin) return;
    const int co = (int)(i % Cout);
    const int k = (int)(i / Cout);
    const int tap = k / Cin, ci = k - tap * Cin;
    dst[i] = ld_any(src, sdt, ((size_t)co * Cin + ci) * 9 + tap);
}

__global__ void pack_vector_kernel(const void* src, int sdt, float* dst, int N, int geglu) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    dst[i] = ld_any(src, sdt, geglu ? geglu_src_row(i, N, geglu) : i);
}

// one wave per output row
__global__ void gemv_kernel(const void* W, int wdt, const void* bias, int bdt, const float* x, float* y, int N, int K,
                            int act) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = x[k];
        if (act == 1) xv = xv / (1.0f + expf(-xv));
        acc = fmaf(ld_any(W, wdt, (size_t)n * K + k), xv, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[n] = acc + (bias ? ld_any(bias, bdt, n) : 0.f);
}

__global__ void add_vec_kernel(const float* a, const float* b, float* o, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) o[i] = a[i] + b[i];
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_kernel(float* out, int dim, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= half) return;
    const float f = expf(-logf(10000.0f) * (float)i / (float)half);
    const float e = (float)t * f;
    out[i] = cosf(e);
    out[half + i] = sinf(e);
}

// the same embedding for `count` float values read from device memory: out[v][dim] (SDXL time_ids)
__global__ void sincos_values_kernel(float* out, int dim, const float* vals, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= half * count) return;
    const int v = i / half, j = i - v * half;
    const float f = expf(-logf(10000.0f) * (float)j / (float)half);
    const float e = vals[v] * f;
    out[(size_t)v * dim + j] = cosf(e);
    out[(size_t)v * dim + half + j] = sinf(e);
}

// x_t = sa*lat + sb*noise (NCHW f32), 3x3 conv (pad 1) to Cout, written twice (CFG halves are
// identical at this point) as token-major [img*2 + cfg][pix][Cout].
// block = 256 threads handles PIX pixels of one image; w is [tap*Cin + ci][Cout] f32.
constexpr int PREP_PIX = 16;
template <typename T>
__global__ __launch_bounds__(256) void prep_conv_in_kernel(const float* __restrict__ lat, const float* __restrict__ noise,
                                                           float sa, float sb, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ out, int Cin,
                                                           int S, int Cout, int dup) {
    extern __shared__ float patch[];                // [PREP_PIX][9*Cin]
    const int img = blockIdx.y, p0 = blockIdx.x * PREP_PIX, K = 9 * Cin, HW = S * S;
    for (int i = threadIdx.x; i < PREP_PIX * K; i += 256) {
        const int pp = i / K, k = i - pp * K;
        const int tap = k / Cin, ci = k - tap * Cin;
        const int pix = p0 + pp;
        float v = 0.f;
        if (pix < HW) {
            const int y = pix / S + tap / 3 - 1, x = pix % S + tap % 3 - 1;
            if ((unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S) {
                const size_t o = (((size_t)img * Cin + ci) * S + y) * S + x;
                v = noise ? sa * lat[o] + sb * noise[o] : lat[o];
            }
        }
        patch[i] = v;
    }
    __syncthreads();
    if (Cout <= 128) {
        // narrow outputs (the VAE's 128 channels): the two halves of the workgroup take half of the pixels each
        // instead of leaving 128 threads idle
        constexpr int HP = PREP_PIX / 2;
        const int co = threadIdx.x % 128, pg = threadIdx.x / 128;
        if (co < Cout) {
            float acc[HP];
            const float bv = bias[co];
#pragma unroll
            for (int pp = 0; pp < HP; ++pp) acc[pp] = bv;
            for (int k = 0; k < K; ++k) {
                const float wv = w[(size_t)k * Cout + co];
#pragma unroll
                for (int pp = 0; pp < HP; ++pp) acc[pp] = fmaf(patch[(pg * HP + pp) * K + k], wv, acc[pp]);
            }
#pragma unroll
            for (int pp = 0; pp < HP; ++pp) {
                const int pix = p0 + pg * HP + pp;
                if (pix < HW) {
                    const T v = (T)acc[pp];
                    for (int d = 0; d < dup; ++d) out[((size_t)(img * dup + d) * HW + pix) * Cout + co] = v;
                }
            }
        }
        return;
    }
    for (int co = threadIdx.x; co < Cout; co += 256) {
        float acc[PREP_PIX];
        const float bv = bias[co];
#pragma unroll
        for (int pp = 0; pp < PREP_PIX; ++pp) acc[pp] = bv;
        for (int k = 0; k < K; ++k) {
            const float wv = w[(size_t)k * Cout + co];
#pragma unroll
            for (int pp = 0; pp < PREP_PIX; ++pp) acc[pp] = fmaf(patch[pp * K + k], wv, acc[pp]);
        }
#pragma unroll
        for (int pp = 0; pp < PREP_PIX; ++pp) {
            const int pix = p0 + pp;
            if (pix < HW) {
                const T v = (T)acc[pp];
                for (int d = 0; d < dup; ++d) out[((size_t)(img * dup + d) * HW + pix) * Cout + co] = v;
            }
        }
    }
}

// The same direct conv with 16-byte stores: a thread owns 8 consecutive output channels of PP pixels (lanes of a pixel
// group cover that pixel's whole channel row, so every store instruction writes full rows), weights come as two float4 per
// tap.  Each output still sums its taps in ascending k order: bit-identical to prep_conv_in_kernel.  Needs Cout % 8 == 0 and
// Cout <= 2048.  The one-channel-per-thread kernel issued a 2-byte store per (pixel, copy) per lane and ran at 1.2 TB/s.
template <typename T, int PP>
__global__ __launch_bounds__(256) void prep_conv_in8_kernel(const float* __restrict__ lat, const float* __restrict__ noise,
                                                            float sa, float sb, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ out, int Cin,
                                                            int S, int Cout, int dup) {
    extern __shared__ float patch[];                // [npg * PP][9*Cin]
    const int CG = Cout / 8, npg = 256 / CG, PIX = npg * PP;
    const int img = blockIdx.y, p0 = blockIdx.x * PIX, K = 9 * Cin, HW = S * S;
    for (int i = threadIdx.x; i < PIX * K; i += 256) {
        const int pp = i / K, k = i - pp * K;
        const int tap = k / Cin, ci = k - tap * Cin;
        const int pix = p0 + pp;
        float v = 0.f;
        if (pix < HW) {
            const int y = pix / S + tap / 3 - 1, x = pix % S + tap % 3 - 1;
            if ((unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S) {
                const size_t o = (((size_t)img * Cin + ci) * S + y) * S + x;
                v = noise ? sa * lat[o] + sb * noise[o] : lat[o];
            }
        }
        patch[i] = v;
    }
    __syncthreads();
    const int cg = threadIdx.x % CG, pg = threadIdx.x / CG;
    if (pg >= npg) return;
    const int co = cg * 8;
    float acc[PP][8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + co), b1 = *reinterpret_cast<const float4*>(bias + co + 4);
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            acc[pp][0] = b0.x; acc[pp][1] = b0.y; acc[pp][2] = b0.z; acc[pp][3] = b0.w;
            acc[pp][4] = b1.x; acc[pp][5] = b1.y; acc[pp][6] = b1.z; acc[pp][7] = b1.w;
        }
    }
    const float* prow = patch + (size_t)pg * PP * K;
    for (int k = 0; k < K; ++k) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)k * Cout + co);
        const float4 w1 = *reinterpret_cast<const float4*>(w + (size_t)k * Cout + co + 4);
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            const float pv = prow[pp * K + k];
            acc[pp][0] = fmaf(pv, w0.x, acc[pp][0]); acc[pp][1] = fmaf(pv, w0.y, acc[pp][1]);
            acc[pp][2] = fmaf(pv, w0.z, acc[pp][2]); acc[pp][3] = fmaf(pv, w0.w, acc[pp][3]);
            acc[pp][4] = fmaf(pv, w1.x, acc[pp][4]); acc[pp][5] = fmaf(pv, w1.y, acc[pp][5]);
            acc[pp][6] = fmaf(pv, w1.z, acc[pp][6]); acc[pp][7] = fmaf(pv, w1.w, acc[pp][7]);
        }
    }
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int pix = p0 + pg * PP + pp;
        if (pix >= HW) continue;
        for (int d = 0; d < dup; ++d) {
            T* o = out + ((size_t)(img * dup + d) * HW + pix) * Cout + co;
            if constexpr (sizeof(T) == 2) {
                typedef T Tx8 __attribute__((ext_vector_type(8)));          // bf16 or fp16
                Tx8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (T)acc[pp][e];
                *reinterpret_cast<Tx8*>(o) = v;
            } else {
                f32x4 v0 = {acc[pp][0], acc[pp][1], acc[pp][2], acc[pp][3]}, v1 = {acc[pp][4], acc[pp][5], acc[pp][6], acc[pp][7]};
                *reinterpret_cast<f32x4*>(o) = v0;
                *reinterpret_cast<f32x4*>(o + 4) = v1;
            }
        }
    }
}

// The VAE's conv_in: 3 -> 128 channels at the full image resolution (512 x 512: 67 MB of output per image; the direct kernels above
// spend their time on per-element index arithmetic and on 2 weight loads per 32 FMAs and reach 1.4 TB/s of it).  A workgroup walks up
// to eight 64-pixel segments of one image row; a thread owns ONE 4-channel quad (its 27 x 4 weights are loaded once and live in
// registers) and 8 consecutive pixels of each segment, whose 3 x 3 x 10 input window it reads once from the staged rows -- the inner
// loop is FMAs only --, and the next segment's input values are in flight under the current segment's FMAs (one segment per
// workgroup was latency-bound: two resident workgroups per CU each waiting out a global and an L2 round trip before 0.9 us of math).
// Same bias-first, ascending-k fmaf chain per output as prep_conv_in_kernel: bit-identical.  STATS: the consumer's GroupNorm
// statistics in the conv epilogue's format (launch_groupnorm_pre: per (64-pixel chunk, 4-channel quad) f32 sum and sum of squares of
// the STORED values, a fixed order: pixels of a thread, the wave's two halves, the four waves) -- a segment IS one chunk.
template <typename T, bool STATS>
__global__ __launch_bounds__(256) void conv_in_rows_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ out, int S, int nseg,
                                                           float* __restrict__ gn_part) {
    constexpr int CIN = 3, COUT = 128, PX = 64, PW = PX + 2, PT = 8, K = 9 * CIN;
    constexpr int NPF = (CIN * 3 * PW + 255) / 256;             // staged input values per thread and segment (3)
    __shared__ float patch[2][CIN * 3][PW + 2];
    __shared__ float s_st[2][4][32][2];
    const int tid = threadIdx.x, cg = tid & 31, pl = tid >> 5;
    const int y = blockIdx.y, im = blockIdx.z, seg0 = blockIdx.x * nseg;
    // the staged window of a segment: value i of the [ci * 3 + dy][66] block (a thread owns i = tid + 256 j); fetched one segment ahead
    int pr[NPF], pxx[NPF];
    const float* prow[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int i = tid + 256 * j, r = i / PW;
        pr[j] = r; pxx[j] = i - r * PW;
        const int ci = r / 3, dy = r - ci * 3, yy = y + dy - 1;
        prow[j] = (i < CIN * 3 * PW && (unsigned)yy < (unsigned)S) ? img + ((size_t)(im * CIN + ci) * S + yy) * S : nullptr;
    }
    float pf[NPF];
    auto fetch = [&](int seg) {
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int x = seg * PX + pxx[j] - 1;
            pf[j] = (prow[j] && (unsigned)x < (unsigned)S) ? prow[j][x] : 0.f;
        }
    };
    fetch(seg0);
    f32x4 wr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wr[k] = *reinterpret_cast<const f32x4*>(w + k * COUT + cg * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + cg * 4);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int s = 0; s < nseg; ++s) {
        const int seg = seg0 + s, buf = s & 1, xb = seg * PX;
#pragma unroll
        for (int j = 0; j < NPF; ++j)
            if (tid + 256 * j < CIN * 3 * PW) patch[buf][pr[j]][pxx[j]] = pf[j];
        __syncthreads();                                        // (also: everybody is done with buffer `buf` of two segments ago)
        if (s + 1 < nseg) fetch(seg + 1);                       // in flight under this segment's FMAs
        float pv[CIN * 3][PT + 2];
#pragma unroll
        for (int r = 0; r < CIN * 3; ++r)
#pragma unroll
            for (int j = 0; j < PT + 2; ++j) pv[r][j] = patch[buf][r][pl * PT + j];
        // (two-wide vectors: v_pk_fma_f32, two FMAs per lane and issue slot)
        f32x2 acc[PT][2];
#pragma unroll
        for (int p = 0; p < PT; ++p) { acc[p][0] = f32x2{bv[0], bv[1]}; acc[p][1] = f32x2{bv[2], bv[3]}; }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const int k = tap * CIN + ci, r = ci * 3 + tap / 3, dx = tap % 3;
                const f32x2 w0 = {wr[k][0], wr[k][1]}, w1 = {wr[k][2], wr[k][3]};
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const f32x2 a = {pv[r][p + dx], pv[r][p + dx]};
                    acc[p][0] = __builtin_elementwise_fma(a, w0, acc[p][0]);
                    acc[p][1] = __builtin_elementwise_fma(a, w1, acc[p][1]);
                }
            }
        T* o = out + ((size_t)im * S * S + (size_t)y * S + xb + pl * PT) * COUT + cg * 4;
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            typedef T Tx4 __attribute__((ext_vector_type(4)));
            Tx4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (T)acc[p][e >> 1][e & 1];
                if (STATS) { const float r = (float)v[e]; ss += r; qq = fmaf(r, r, qq); }
            }
            *reinterpret_cast<Tx4*>(o + (size_t)p * COUT) = v;
        }
        if constexpr (STATS) {
            ss += __shfl_xor(ss, 32, 64);
            qq += __shfl_xor(qq, 32, 64);
            if ((tid & 63) < 32) { s_st[buf][tid >> 6][cg][0] = ss; s_st[buf][tid >> 6][cg][1] = qq; }
            __syncthreads();
            if (tid < 32) {
                const float a = ((s_st[buf][0][cg][0] + s_st[buf][1][cg][0]) + s_st[buf][2][cg][0]) + s_st[buf][3][cg][0];
                const float q = ((s_st[buf][0][cg][1] + s_st[buf][1][cg][1]) + s_st[buf][2][cg][1]) + s_st[buf][3][cg][1];
                const size_t chunk = (size_t)im * (S * S / PX) + ((size_t)y * S + xb) / PX;
                *reinterpret_cast<float2*>(gn_part + (chunk * (COUT / 4) + cg) * 2) = make_float2(a, q);
            }
        }
    }
}

__global__ void convert_kernel(const float* src, void* dst, int ddt, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st_any(dst, ddt, i, src[i]);
}

}  // namespace

int pack_linear(const void* src, int sdt, void* dst, int ddt, int N, int K, int geglu, hipStream_t s) {
    if (geglu && ((geglu != 16 && geglu != 32) || N % (2 * geglu))) return DSIM_ERR_INVALID;
    const size_t n = (size_t)N * K;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, sdt, dst, ddt, N, K, geglu);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int pack_conv3(const void* src, int sdt, void* dst, int ddt, int Cout, int Cin, hipStream_t s) {
    const size_t n = (size_t)Cout * 9 * Cin;
    hipLaunchKernelGGL(pack_conv3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, sdt, dst, ddt, Cout, Cin);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int pack_conv_in(const void* src, int sdt, float* dst, int Cout, int Cin, hipStream_t s) {
    const size_t n = (size_t)Cout * 9 * Cin;
    hipLaunchKernelGGL(pack_conv_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, sdt, dst, Cout, Cin);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int pack_vector(const void* src, int sdt, float* dst, int N, int geglu, hipStream_t s) {
    hipLaunchKernelGGL(pack_vector_kernel, dim3((N + 255) / 256), dim3(256), 0, s, src, sdt, dst, N, geglu);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int gemv_f32(const void* W, int wdt, const void* bias, int bdt, const float* x, float* y, int N, int K, int act,
             hipStream_t s) {
    hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, wdt, bias, bdt, x, y, N, K, act);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int add_vectors_f32(const float* a, const float* b, float* out, int N, hipStream_t s) {
    hipLaunchKernelGGL(add_vec_kernel, dim3((N + 255) / 256), dim3(256), 0, s, a, b, out, N);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int timestep_sincos(float* out, int dim, int t, hipStream_t s) {
    hipLaunchKernelGGL(timestep_kernel, dim3((dim / 2 + 255) / 256), dim3(256), 0, s, out, dim, t);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int sincos_values(float* out, int dim, const float* vals, int count, hipStream_t s) {
    hipLaunchKernelGGL(sincos_values_kernel, dim3((dim / 2 * count + 255) / 256), dim3(256), 0, s, out, dim, vals, count);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
bool conv_in_rows_applies(int Cin, int S, int Cout) { return Cin == 3 && Cout == 128 && S % 64 == 0 && S <= 32768; }
int conv_in_rows(const float* images, const float* w, const float* bias, void* out, int dtype, int n_img, int S, float* gn_part,
                 hipStream_t st) {
    if (n_img < 1 || n_img > 65535 || (gn_part && dtype == DSIM_F32)) return DSIM_ERR_INVALID;
    int nseg = 8;                                         // 64-pixel segments of an image row one workgroup walks (weights loaded once)
    while ((S / 64) % nseg) nseg >>= 1;
    const dim3 grid(S / 64 / nseg, S, n_img);
    if (dtype == DSIM_BF16) {
        if (gn_part) hipLaunchKernelGGL((conv_in_rows_kernel<bf16_t, true>), grid, dim3(256), 0, st, images, w, bias, (bf16_t*)out, S, nseg, gn_part);
        else hipLaunchKernelGGL((conv_in_rows_kernel<bf16_t, false>), grid, dim3(256), 0, st, images, w, bias, (bf16_t*)out, S, nseg, gn_part);
    } else if (dtype == DSIM_F16) {
        if (gn_part) hipLaunchKernelGGL((conv_in_rows_kernel<f16_t, true>), grid, dim3(256), 0, st, images, w, bias, (f16_t*)out, S, nseg, gn_part);
        else hipLaunchKernelGGL((conv_in_rows_kernel<f16_t, false>), grid, dim3(256), 0, st, images, w, bias, (f16_t*)out, S, nseg, gn_part);
    } else if (dtype == DSIM_F32) {
        hipLaunchKernelGGL((conv_in_rows_kernel<float, false>), grid, dim3(256), 0, st, images, w, bias, (float*)out, S, nseg, gn_part);
    } else {
        return DSIM_ERR_INVALID;
    }
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
int prep_conv_in(const float* lat, const float* noise, float sa, float sb, const float* w, const float* bias, void* out,
                 int dtype, int n_img, int Cin, int S, int Cout, int dup, hipStream_t st) {
    if (g_prep8 && Cout % 8 == 0 && Cout / 8 <= 256) {
        constexpr int PP = 4;
        const int PIX = (256 / (Cout / 8)) * PP;
        const dim3 grid8((S * S + PIX - 1) / PIX, n_img);
        const size_t lds8 = (size_t)PIX * 9 * Cin * sizeof(float);
        if (lds8 <= 48 * 1024) {
            if (dtype == DSIM_BF16)
                hipLaunchKernelGGL((prep_conv_in8_kernel<bf16_t, PP>), grid8, dim3(256), lds8, st, lat, noise, sa, sb, w, bias, (bf16_t*)out, Cin, S, Cout, dup);
            else if (dtype == DSIM_F16)
                hipLaunchKernelGGL((prep_conv_in8_kernel<f16_t, PP>), grid8, dim3(256), lds8, st, lat, noise, sa, sb, w, bias, (f16_t*)out, Cin, S, Cout, dup);
            else if (dtype == DSIM_F32)
                hipLaunchKernelGGL((prep_conv_in8_kernel<float, PP>), grid8, dim3(256), lds8, st, lat, noise, sa, sb, w, bias, (float*)out, Cin, S, Cout, dup);
            else
                return DSIM_ERR_INVALID;
            DSIM_HIP_CHECK(hipGetLastError());
            return DSIM_OK;
        }
    }
    const dim3 grid((S * S + PREP_PIX - 1) / PREP_PIX, n_img), block(256);
    const size_t lds = (size_t)PREP_PIX * 9 * Cin * sizeof(float);
    if (dtype == DSIM_BF16)
        hipLaunchKernelGGL(prep_conv_in_kernel<bf16_t>, grid, block, lds, st, lat, noise, sa, sb, w, bias, (bf16_t*)out, Cin, S, Cout, dup);
    else if (dtype == DSIM_F16)
        hipLaunchKernelGGL(prep_conv_in_kernel<f16_t>, grid, block, lds, st, lat, noise, sa, sb, w, bias, (f16_t*)out, Cin, S, Cout, dup);
    else if (dtype == DSIM_F32)
        hipLaunchKernelGGL(prep_conv_in_kernel<float>, grid, block, lds, st, lat, noise, sa, sb, w, bias, (float*)out, Cin, S, Cout, dup);
    else
        return DSIM_ERR_INVALID;
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}
// out[(b * 2 + c)][i] = in[b][i] for c in {0, 1}: one batch element of `per` 16-byte chunks becomes its two CFG copies
__global__ void dup_batch_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t per, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / per, r = i - b * per;
    const u32x4 v = in[i];
    out[(2 * b) * per + r] = v;
    out[(2 * b + 1) * per + r] = v;
}

int dup_batch(const void* in, void* out, int n_batch, size_t bytes_per_elem, hipStream_t s) {
    if (bytes_per_elem % 16) return DSIM_ERR_INVALID;
    const size_t per = bytes_per_elem / 16, total = per * (size_t)n_batch;
    hipLaunchKernelGGL(dup_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const u32x4*)in, (u32x4*)out, per, total);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

namespace {
__global__ void resize_nearest_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int Hin, int Win, int Hout, int Wout,
                                      int cpr /*16-byte chunks per pixel*/, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cpr);
    const size_t pix = i / cpr;
    const int ox = (int)(pix % Wout), oy = (int)((pix / Wout) % Hout);
    const size_t b = pix / ((size_t)Wout * Hout);
    // torch's nearest: scale = (float)in / out, src = min((int)floorf(dst * scale), in - 1)
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
    const int iy = min((int)floorf((float)oy * sy), Hin - 1), ix = min((int)floorf((float)ox * sx), Win - 1);
    out[i] = in[((b * Hin + iy) * Win + ix) * cpr + c];
}
}  // namespace
int resize_nearest(const void* in, void* out, int B, int Hin, int Win, int Hout, int Wout, size_t row_bytes, hipStream_t s) {
    if (row_bytes % 16 || B < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return DSIM_ERR_INVALID;
    const int cpr = (int)(row_bytes / 16);
    const size_t total = (size_t)B * Hout * Wout * cpr;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const u32x4*)in, (u32x4*)out, Hin, Win,
                       Hout, Wout, cpr, total);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

int convert_f32_to(const float* src, void* dst, int dtype, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, dtype, n);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

// ---- host-adjacent arithmetic of the path, kept on the device so that decoded pixels are the only thing the host produces ----
namespace {

// process_image after the resize (/root/reference/diffsim/diffsim.py:31-41): uint8 HWC -> /255 -> (x - 0.5) / 0.5 -> NCHW.
// Plain IEEE f32 operations in numpy's order, so the result is bit-identical to the host path (asserted against golden G1);
// half = 1: the SD1.5 pipeline's fp16 image cast (diffsim.py:93) folded in, stored as f32 values that are exactly fp16.
__global__ void image_preprocess_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int HW, size_t total, int half) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per pixel
    if (i >= total) return;
    const size_t n = i / HW, pix = i - n * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = __fdiv_rn((float)src[i * 3 + c], 255.0f);
        v = __fdiv_rn(__fsub_rn(v, 0.5f), 0.5f);
        asm volatile("" : "+v"(v));              // (f32 result first, then the fp16 cast: no fused convert)
        if (half) v = (float)(_Float16)v;
        dst[(n * 3 + c) * HW + pix] = v;
    }
}

// DiagonalGaussianDistribution.sample + scaling (diffusers AutoencoderKL as prepare_image_latents uses it, diffsim.py:92-96):
// lat = sf * (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps); eps is one draw shared by every image (eps_n = 1) or per image;
// round16 = 1 rounds the result through fp16 (the fp16 pipeline's latents, diffsim_xl.py:63 / noise_dtype = float16).
// moments [n][2C][hw]; image j of the batch takes every `stride`-th image starting at `first` (triplets: ref / left / right).
__global__ void latent_sample_kernel(const float* __restrict__ mom, const float* __restrict__ eps, float* __restrict__ out, int C,
                                     int hw, int n_out, int first, int stride, int eps_n, float sf, int round16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = C * hw;
    if (i >= n_out * per) return;
    const int j = i / per, r = i - j * per;
    const float* m = mom + (size_t)(first + j * stride) * 2 * per;
    const float mean = m[r];
    float lv = m[per + r];
    lv = fminf(fmaxf(lv, -30.0f), 20.0f);
    const float sd = expf(__fmul_rn(0.5f, lv));
    // separately rounded multiply / add / multiply, as the elementwise tensor expression sf * (mean + std * eps) evaluates
    // (no fused multiply-add): bit-identical to the host-framework form it replaces
    float v = __fmul_rn(sf, __fadd_rn(mean, __fmul_rn(sd, eps[(eps_n > 1 ? (size_t)j * per : 0) + r])));
    // the f32 product is rounded FIRST, then to fp16 (two roundings, as latents.to(float16) does): the opaque copy keeps hipcc
    // from folding the multiply and the conversion into one v_fma_mix, which rounds the exact product once and lands on the
    // other side of an fp16 tie
    asm volatile("" : "+v"(v));
    if (round16) v = (float)(_Float16)v;
    out[i] = v;
}

}  // namespace

int image_preprocess(const unsigned char* hwc, float* out, int n, int H, int W, int half, hipStream_t s) {
    if (!hwc || !out || n < 1 || H < 1 || W < 1) return DSIM_ERR_INVALID;
    const size_t total = (size_t)n * H * W;
    hipLaunchKernelGGL(image_preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hwc, out, H * W, total, half);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

int latent_sample(const float* moments, const float* eps, float* out, int n_out, int first, int stride, int C, int hw, int eps_n,
                  float sf, int round16, hipStream_t s) {
    if (!moments || !eps || !out || n_out < 1 || C < 1 || hw < 1 || stride < 1 || first < 0) return DSIM_ERR_INVALID;
    const int total = n_out * C * hw;
    hipLaunchKernelGGL(latent_sample_kernel, dim3((total + 255) / 256), dim3(256), 0, s, moments, eps, out, C, hw, n_out, first, stride,
                       eps_n, sf, round16);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

}  // namespace dsim
