// Development micro-benchmark (gfx950): the instruction mix of ONE pipeline step of attn_long_kernel<40> (attention.hip) in its
// current form and in the "mixed MFMA shape" form the round-3 review asked to be measured before it is built:
//   current: QK^T 3 x v_mfma_f32_32x32x16_bf16 (d 40 -> 48)  +  PV 4 x 32x32x16 (d 41 -> 64)       = 224 MFMA cycles per unit
//   mixed  : QK^T 3 x 32x32x16                               +  PV 6 x v_mfma_f32_16x16x32_bf16     = 192 MFMA cycles per unit
//            (3 d-tiles of 16 rows x 2 query tiles; the P fragments re-laid by 4 v_permlane16_swap per unit)
// both with the unit's softmax between the MFMAs exactly as the kernel places it: 16 v_exp_f32 + 8 v_cvt_pk per lane, three
// exponentials per MFMA group.  Operands live in registers (no LDS, no global memory): this is the ISSUE bound of the two step
// bodies at the kernel's occupancy (two 256-thread workgroups per CU = two waves per SIMD), on random data, i.e. an upper bound
// on what the rewrite can gain.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_attn_mix.hip -o tools/ubench_attn_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(x & 0xffff) / 32768.0f - 1.0f;
}

// MIX = 0: current step body; 1: mixed shapes
template <int MIX>
__global__ __launch_bounds__(256, 2) void step_body(float* out, int iters) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 kf[3], qf[3], vf[4], vf16[3];
    f32x16 s, sprev, o[2];
    f32x4 o16[3][2];
    bf16x8 p[2];
    for (int i = 0; i < 3; ++i)
        for (int e = 0; e < 8; ++e) { kf[i][e] = (__bf16)rnd(tid * 31 + i * 8 + e); qf[i][e] = (__bf16)(0.3f * rnd(tid * 17 + i * 8 + e + 99)); }
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) vf[i][e] = (__bf16)rnd(tid * 13 + i * 8 + e + 7);
    for (int i = 0; i < 3; ++i)
        for (int e = 0; e < 8; ++e) vf16[i][e] = (__bf16)rnd(tid * 11 + i * 8 + e + 5);
    for (int r = 0; r < 16; ++r) { sprev[r] = -1.0f - 0.1f * r; o[0][r] = o[1][r] = 0.f; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) o16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < 8; ++e) p[0][e] = p[1][e] = (__bf16)0.25f;
    for (int it = 0; it < iters; ++it) {
        // one unit: QK of the NEXT unit (-> s), the softmax of the PREVIOUS QK result (sprev -> p), PV of the converted p
        constexpr int NM = MIX ? 9 : 7;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            if (g < 3) {
                if (g == 0) { f32x16 z; for (int r = 0; r < 16; ++r) z[r] = -2.0f; s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], z, 0, 0, 0); }
                else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g], qf[g], s, 0, 0, 0);
            } else if (!MIX) {
                const int s2 = (g - 3) / 2, db = (g - 3) % 2;
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2 * 2 + db], p[s2], o[db], 0, 0, 0);
            } else {
                const int dt = (g - 3) / 2, qt = (g - 3) % 2;
                o16[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf16[dt], p[qt], o16[dt][qt], 0, 0, 0);
            }
            // three exponentials per MFMA group over the first groups (the kernel's EPG = 3), the conversions behind their eight
            constexpr int EPG = MIX ? 2 : 3;
#pragma unroll
            for (int r = g * EPG; r < (g + 1) * EPG && r < 16; ++r) sprev[r] = __builtin_amdgcn_exp2f(sprev[r]);
            if (g == (MIX ? 4 : 3)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) p[0][e] = (__bf16)sprev[e];
            }
            if (g == NM - 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) p[1][e] = (__bf16)sprev[8 + e];
                if (MIX) {          // re-lay the two P fragments for the 16-wide query tiles: one swap per register pair
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 a = __builtin_bit_cast(u32x4, p[0]), b = __builtin_bit_cast(u32x4, p[1]);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const auto r2 = __builtin_amdgcn_permlane16_swap(a[w], b[w], false, false);
                        a[w] = r2[0]; b[w] = r2[1];
                    }
                    p[0] = __builtin_bit_cast(bf16x8, a); p[1] = __builtin_bit_cast(bf16x8, b);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // next unit's softmax input: this unit's scores, pulled back into the exponent's useful range
#pragma unroll
        for (int r = 0; r < 16; ++r) sprev[r] = -1.0f - 1e-3f * s[r];
    }
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += o[0][r] + o[1][r] + sprev[r];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) acc += o16[i][j][r];
    out[tid] = acc;
}

template <int MIX>
static float run(float* out, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(step_body<MIX>, dim3(512), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(step_body<MIX>, dim3(512), dim3(256), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    const int iters = 200000;
    std::vector<float> m0, m1;
    for (int r = 0; r < 7; ++r) { m0.push_back(run<0>(out, iters)); m1.push_back(run<1>(out, iters)); }     // interleaved rounds
    std::sort(m0.begin(), m0.end()); std::sort(m1.begin(), m1.end());
    // 512 workgroups on 256 CUs = 2 per CU, 2 waves per SIMD: each SIMD runs 2 x iters units
    const double us0 = m0[3] * 1e3, us1 = m1[3] * 1e3;
    printf("step body, %d units per wave, 2 waves per SIMD, median of 7 interleaved rounds\n", iters);
    printf("  current (3 + 4 x 32x32x16)            : %8.2f ms  = %6.1f ns per unit-pair per SIMD\n", m0[3], us0 * 1e3 / iters);
    printf("  mixed   (3 x 32x32x16 + 6 x 16x16x32) : %8.2f ms  = %6.1f ns per unit-pair per SIMD   ratio %.3f\n", m1[3], us1 * 1e3 / iters, m1[3] / m0[3]);
    return 0;
}
