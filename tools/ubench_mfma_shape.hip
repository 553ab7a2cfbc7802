// Development micro-benchmark (gfx950): does the bf16 MFMA SHAPE change the wall time of an LDS-fed GEMM inner loop?
// MI355X_MICROARCH.md "DVFS give-back" item 7 reports v_mfma_f32_16x16x32_bf16 loops at ~1.12-1.15x the FLOP/s of
// v_mfma_f32_32x32x16_bf16 loops at equal cycles per FLOP, because the chip holds a higher clock on the small shape.
// This measures it in the regime of gemm.hip's K loop: 8 waves per workgroup (2 per SIMD), one workgroup per CU, a 64 x 160
// output tile per wave, every operand fragment re-read from LDS by ds_read_b128 (0.7 reads per 32x32x16 MFMA), random data,
// fragments double-buffered with the read -> MFMA order pinned as in gemm.hip.  Reports wall time, TF/s and the in-kernel clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_shape.hip -o tools/ubench_mfma_shape ; run: ./tools/ubench_mfma_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LDS_BYTES = 144 * 1024;

// SHAPE 0: 32x32x16, per 16-deep step 2 A-side + 5 B-side fragments -> 10 MFMAs (64 x 160 per wave)
// SHAPE 1: 16x16x32, per 32-deep step 4 + 10 fragments -> 40 MFMAs (same tile, same LDS bytes per FLOP)
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void loop_kernel(const unsigned* __restrict__ seed, float* __restrict__ out, long long* __restrict__ clk,
                                                      int ksteps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // random bf16 bit patterns with a sane exponent range
    for (int i = tid; i < LDS_BYTES / 4; i += 512) {
        unsigned x = seed[(blockIdx.x * 7919 + i) & 65535] * 2654435761u + i;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        const unsigned lo = 0x3f00u | (x & 0x80ffu), hi = 0x3f00u | ((x >> 16) & 0x80ffu);
        reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
    }
    __syncthreads();
    const char* base = smem + (wave & 3) * 8192 + lane * 16;      // conflict-free: a wave reads 1 KB contiguous per fragment
    float s = 0.f;
    long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if constexpr (SHAPE == 0) {
        f32x16 acc[2][5];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        bf16x8 fa[2][2], fb[2][5];
        auto ld = [&](int k, int set) {
            const char* p = base + ((k * 7) & 63) * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
#pragma unroll
            for (int j = 0; j < 5; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(p + (2 + j) * 1024);
        };
        ld(0, 0);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < ksteps; k += 2) {
            ld(k + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ld(k + 2, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        f32x4 acc[4][10];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        // per 32-deep step: the four A-side fragments are held, the ten B-side fragments stream through a ring of 4 in two halves
        bf16x8 fa[2][4], fb[2][5];
        auto lda = [&](int k, int set) {
            const char* p = base + ((k * 14) & 63) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
        };
        auto ldb = [&](int k, int h, int set) {
            const char* p = base + ((k * 14 + 4 + 5 * h) & 63) * 1024;
#pragma unroll
            for (int j = 0; j < 5; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(p + j * 1024);
        };
        lda(0, 0); ldb(0, 0, 0);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < ksteps; k += 2) {     // one trip = two 32-deep steps = 4 x (16-deep steps): same depth per trip as 4 trips above
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ldb(k + u, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[u][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                lda(k + u + 1, u ^ 1); ldb(k + u + 1, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][5 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[u][i], acc[i][5 + j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) { clk[(blockIdx.x * 8 + wave) * 2] = t1 - t0; clk[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const int nb = 256;
    unsigned* seed; float* out; long long* clk;
    HC(hipMalloc(&seed, 65536 * 4)); HC(hipMalloc(&out, nb * 512 * 4)); HC(hipMalloc(&clk, nb * 8 * 2 * 8));
    std::vector<unsigned> hs(65536);
    for (int i = 0; i < 65536; ++i) hs[i] = i * 747796405u + 2891336453u;
    HC(hipMemcpy(seed, hs.data(), 65536 * 4, hipMemcpyHostToDevice));
    HC(hipFuncSetAttribute((const void*)loop_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    HC(hipFuncSetAttribute((const void*)loop_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t a, b; HC(hipEventCreate(&a)); HC(hipEventCreate(&b));
    // shape 0: ksteps 16-deep steps; shape 1: ksteps/2 32-deep steps -> equal FLOPs: per wave 64 x 160 x 16 x ksteps x 2
    const int ks0 = 16384;
    const double flop = 2.0 * 64 * 160 * 16.0 * ks0 * 8 * nb;
    std::vector<float> ms[2];
    std::vector<double> ghz[2];
    for (int r = 0; r < 7; ++r)
        for (int sh = 0; sh < 2; ++sh) {
            HC(hipEventRecord(a, 0));
            if (sh == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(nb), dim3(512), LDS_BYTES, 0, seed, out, clk, ks0);
            else hipLaunchKernelGGL(loop_kernel<1>, dim3(nb), dim3(512), LDS_BYTES, 0, seed, out, clk, ks0 / 2);
            HC(hipEventRecord(b, 0)); HC(hipEventSynchronize(b));
            float t; HC(hipEventElapsedTime(&t, a, b));
            std::vector<long long> hc(nb * 8 * 2);
            HC(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> g;
            for (int i = 0; i < nb * 8; ++i) g.push_back((double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1);   // cycles per 10 ns tick -> GHz
            std::sort(g.begin(), g.end());
            if (r) { ms[sh].push_back(t); ghz[sh].push_back(g[g.size() / 2]); }
        }
    for (int sh = 0; sh < 2; ++sh) {
        std::sort(ms[sh].begin(), ms[sh].end()); std::sort(ghz[sh].begin(), ghz[sh].end());
        printf("%s  min %.3f ms  median %.3f ms  %.1f TF/s (median)  in-kernel clock %.3f GHz  cycles/MFMA-equivalent(32x32x16) %.1f\n",
               sh ? "16x16x32" : "32x32x16", ms[sh][0], ms[sh][3], flop / ms[sh][3] / 1e9, ghz[sh][3],
               ms[sh][3] * 1e-3 * ghz[sh][3] * 1e9 / (10.0 * ks0 * 2));
    }
    return 0;
}
