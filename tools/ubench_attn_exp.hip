// Development micro-benchmark (gfx950): the softmax exponentials of ONE pipeline step of attn_long_kernel<40> (attention.hip),
// measured as the round-4 review asked before touching the kernel: the unit's 16 v_exp_f32 per lane (the quarter-rate transcendental
// unit, 8 issue cycles each) against forms that move NP of them onto the FMA lanes --
//   poly  : exp2 by Cody-Waite, n = floor(x), f = x - n, degree-3 polynomial on [0, 1) (|rel err| < 1.1e-4 < 2^-9: P is rounded to
//           bf16 anyway), v_ldexp_f32: 7 plain VALU per element
//   ppoly : the same on TWO elements at once with v_pk_add_f32 / v_pk_fma_f32 where a packed form exists: 10 VALU per pair
// placed between the step's 7 MFMAs (3 QK^T + 4 PV, 32x32x16) exactly as the kernel places its v_exp (three per MFMA group), the
// v_cvt_pk conversions behind them; operands in registers, two waves per SIMD (the kernel's occupancy), random data, interleaved rounds.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_attn_exp.hip -o tools/ubench_attn_exp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(x & 0xffff) / 32768.0f - 1.0f;
}
// 2^x, x <= 0 (scores relative to the row's reference): degree-3 minimax of 2^f on [0, 1), |rel err| < 1.1e-4
__device__ __forceinline__ float exp2_poly(float x) {
    const float n = __builtin_floorf(x), f = x - n;
    float p = __builtin_fmaf(0.0790204f, f, 0.2242f);
    p = __builtin_fmaf(p, f, 0.6967f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}
__device__ __forceinline__ f32x2 exp2_ppoly(f32x2 x) {
    const f32x2 n = {__builtin_floorf(x[0]), __builtin_floorf(x[1])};
    const f32x2 f = x - n;                                   // v_pk_add_f32
    f32x2 p = __builtin_elementwise_fma((f32x2){0.0790204f, 0.0790204f}, f, (f32x2){0.2242f, 0.2242f});      // v_pk_fma_f32
    p = __builtin_elementwise_fma(p, f, (f32x2){0.6967f, 0.6967f});
    p = __builtin_elementwise_fma(p, f, (f32x2){1.0f, 1.0f});
    return (f32x2){__builtin_ldexpf(p[0], (int)n[0]), __builtin_ldexpf(p[1], (int)n[1])};
}

// NP of the unit's 16 exponentials by polynomial (the LAST NP: they then sit in the later MFMA groups); PK: the packed form
template <int NP, bool PK>
__global__ __launch_bounds__(256, 2) void step_body(float* out, int iters) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 kf[3], qf[3], vf[4];
    f32x16 s, sprev, o[2];
    bf16x8 p[2];
    for (int i = 0; i < 3; ++i)
        for (int e = 0; e < 8; ++e) { kf[i][e] = (__bf16)rnd(tid * 31 + i * 8 + e); qf[i][e] = (__bf16)(0.3f * rnd(tid * 17 + i * 8 + e + 99)); }
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) vf[i][e] = (__bf16)rnd(tid * 13 + i * 8 + e + 7);
    for (int r = 0; r < 16; ++r) { sprev[r] = -1.0f - 0.1f * r; o[0][r] = o[1][r] = 0.f; }
    for (int e = 0; e < 8; ++e) p[0][e] = p[1][e] = (__bf16)0.25f;
    for (int it = 0; it < iters; ++it) {
        constexpr int NM = 7, EPG = 3;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            if (g < 3) {
                if (g == 0) { f32x16 z; for (int r = 0; r < 16; ++r) z[r] = -2.0f; s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], z, 0, 0, 0); }
                else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g], qf[g], s, 0, 0, 0);
            } else {
                const int s2 = (g - 3) / 2, db = (g - 3) % 2;
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2 * 2 + db], p[s2], o[db], 0, 0, 0);
            }
#pragma unroll
            for (int r = g * EPG; r < (g + 1) * EPG && r < 16; ++r) {
                if (r < 16 - NP) sprev[r] = __builtin_amdgcn_exp2f(sprev[r]);
                else if (!PK) sprev[r] = exp2_poly(sprev[r]);
                else if (((r - (16 - NP)) & 1) == 0 && r + 1 < 16) {          // a pair: elements r, r + 1 (the second one is skipped below)
                    const f32x2 e2 = exp2_ppoly((f32x2){sprev[r], sprev[r + 1]});
                    sprev[r] = e2[0]; sprev[r + 1] = e2[1];
                }
            }
            if (g == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) p[0][e] = (__bf16)sprev[e];
            }
            if (g == NM - 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) p[1][e] = (__bf16)sprev[8 + e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sprev[r] = -1.0f - 1e-3f * s[r];
    }
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += o[0][r] + o[1][r] + sprev[r];
    out[tid] = acc;
}

template <int NP, bool PK>
static float run(float* out, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto k = step_body<NP, PK>;
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    const int iters = 100000;
    std::vector<float> m[7];
    for (int r = 0; r < 7; ++r) {          // interleaved rounds
        m[0].push_back(run<0, false>(out, iters)); m[1].push_back(run<2, false>(out, iters)); m[2].push_back(run<4, false>(out, iters));
        m[3].push_back(run<8, false>(out, iters)); m[4].push_back(run<2, true>(out, iters)); m[5].push_back(run<4, true>(out, iters));
        m[6].push_back(run<8, true>(out, iters));
    }
    const char* nm[7] = {"16 v_exp (shipped)", "14 v_exp + 2 poly", "12 v_exp + 4 poly", " 8 v_exp + 8 poly", "14 v_exp + 2 packed poly",
                         "12 v_exp + 4 packed poly", " 8 v_exp + 8 packed poly"};
    printf("attn_long step body, %d units per wave, 2 waves per SIMD, median of 7 interleaved rounds\n", iters);
    for (int k = 0; k < 7; ++k) {
        std::sort(m[k].begin(), m[k].end());
        printf("  %-26s %8.2f ms  = %6.1f ns per unit-pair per SIMD   ratio %.3f\n", nm[k], m[k][3], m[k][3] * 1e6 / iters, m[k][3] / m[0][3]);
    }
    return 0;
}
