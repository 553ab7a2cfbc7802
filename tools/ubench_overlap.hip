// Development micro-benchmark (gfx950): how do MFMA and VALU issue overlap on one SIMD
//   (a) inside ONE wave's instruction stream (independent VALU placed between MFMAs), and
//   (b) ACROSS two waves of one SIMD (one wave MFMA-only, its partner VALU-only)?
// The attention kernel's schedule (attention.hip) is designed from these numbers; profiles/r02_ubench_overlap.txt
// keeps the output.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o tools/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- (a) one wave: per iteration NM MFMAs, each followed by NE v_exp_f32 and NF v_fma_f32 (all independent) ----
template <int NM, int NE, int NF>
__global__ __launch_bounds__(256, 1) void in_wave(float* out, long long* cyc, int iters) {
    f32x16 acc[4];
    float e[8], f[8];
    bf16x8 a, b;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) { e[i] = -0.01f * (threadIdx.x + i); f[i] = 0.5f + i; a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NE; ++i) e[(m * NE + i) & 7] = __builtin_amdgcn_exp2f(e[(m * NE + i) & 7]);
#pragma unroll
            for (int i = 0; i < NF; ++i) f[(m * NF + i) & 7] = __builtin_fmaf(f[(m * NF + i) & 7], 0.999f, 0.001f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += e[i] + f[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- (b) 8 waves per workgroup = 2 per SIMD: waves 0-3 run ROLE_A, waves 4-7 ROLE_B --------------------------
// role 0: idle (exit at once), 1: MFMA only (NM per iteration), 2: exp only (NE), 3: fma only (NF), 4: both MFMA + exp interleaved
template <int RA, int RB, int NM, int NE, int NF>
__global__ __launch_bounds__(512, 2) void two_waves(float* out, long long* cyc, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? RA : RB;
    f32x16 acc[4];
    float e[8], f[8];
    bf16x8 a, b;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) { e[i] = -0.01f * (threadIdx.x + i); f[i] = 0.5f + i; a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NE; ++i) e[i & 7] = __builtin_amdgcn_exp2f(e[i & 7]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NF; ++i) f[i & 7] = __builtin_fmaf(f[i & 7], 0.999f, 0.001f);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NE / NM; ++i) e[(m * (NE / NM) + i) & 7] = __builtin_amdgcn_exp2f(e[(m * (NE / NM) + i) & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += e[i] + f[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* d_out;
static long long* d_cyc;

template <int NM, int NE, int NF>
void run_in_wave() {
    const int iters = 4000, wgs = 256;
    hipLaunchKernelGGL((in_wave<NM, NE, NF>), dim3(wgs), dim3(256), 0, 0, d_out, d_cyc, 10);
    hipLaunchKernelGGL((in_wave<NM, NE, NF>), dim3(wgs), dim3(256), 0, 0, d_out, d_cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> c(wgs * 4);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per = (double)c[c.size() / 2] / iters / NM;
    printf("in-wave  1 wave/SIMD  per MFMA: +%d exp +%d fma : %7.1f cycles per MFMA gap\n", NE, NF, per);
}

template <int RA, int RB, int NM, int NE, int NF>
void run_two(const char* what) {
    const int iters = 4000, wgs = 256;
    hipLaunchKernelGGL((two_waves<RA, RB, NM, NE, NF>), dim3(wgs), dim3(512), 0, 0, d_out, d_cyc, 10);
    hipLaunchKernelGGL((two_waves<RA, RB, NM, NE, NF>), dim3(wgs), dim3(512), 0, 0, d_out, d_cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> c(wgs * 8);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> a, b;
    for (int w = 0; w < wgs; ++w) for (int i = 0; i < 8; ++i) (i < 4 ? a : b).push_back(c[w * 8 + i]);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("two-wave %-34s waves0-3: %8.1f cyc/iter   waves4-7: %8.1f cyc/iter   (per iter: %d MFMA / %d exp / %d fma)\n", what,
           (double)a[a.size() / 2] / iters, (double)b[b.size() / 2] / iters, NM, NE, NF);
}

int main() {
    hipMalloc(&d_out, 256 * 512 * 4);
    hipMalloc(&d_cyc, 256 * 8 * 8);
    run_in_wave<4, 0, 0>();
    run_in_wave<4, 1, 0>();
    run_in_wave<4, 2, 0>();
    run_in_wave<4, 3, 0>();
    run_in_wave<4, 4, 0>();
    run_in_wave<4, 0, 2>();
    run_in_wave<4, 0, 4>();
    run_in_wave<4, 0, 6>();
    run_in_wave<4, 0, 8>();
    run_in_wave<4, 2, 2>();
    run_in_wave<4, 2, 4>();
    run_in_wave<4, 3, 2>();
    run_two<1, 0, 8, 16, 32>("MFMA | idle");
    run_two<2, 0, 8, 16, 32>("exp  | idle");
    run_two<3, 0, 8, 16, 32>("fma  | idle");
    run_two<1, 1, 8, 16, 32>("MFMA | MFMA");
    run_two<2, 2, 8, 16, 32>("exp  | exp");
    run_two<3, 3, 8, 16, 32>("fma  | fma");
    run_two<1, 2, 8, 16, 32>("MFMA | exp   (8 MFMA vs 16 exp)");
    run_two<1, 2, 8, 32, 32>("MFMA | exp   (8 MFMA vs 32 exp)");
    run_two<1, 3, 8, 16, 32>("MFMA | fma   (8 MFMA vs 32 fma)");
    run_two<1, 3, 8, 16, 64>("MFMA | fma   (8 MFMA vs 64 fma)");
    run_two<4, 4, 8, 16, 32>("MFMA+2exp interleaved | same");
    run_two<4, 0, 8, 16, 32>("MFMA+2exp interleaved | idle");
    run_two<4, 4, 8, 24, 32>("MFMA+3exp interleaved | same");
    run_two<4, 0, 8, 24, 32>("MFMA+3exp interleaved | idle");
    return 0;
}
