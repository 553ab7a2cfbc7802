// Development micro-benchmark: do v_exp_f32 (transcendental), plain VALU (v_fma_f32) and MFMA overlap on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NEXP, int NFMA, int NMFMA>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float e[8], f[8];
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { e[i] = -0.001f * (threadIdx.x + i); f[i] = 0.5f + i; a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NEXP; ++i) e[i & 7] = __builtin_amdgcn_exp2f(e[i & 7]) - 1.0001f;
#pragma unroll
            for (int i = 0; i < NFMA; ++i) f[i & 7] = __builtin_fmaf(f[i & 7], 0.999f, 0.001f);
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += e[i] + f[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NEXP, int NFMA, int NMFMA>
void run(const char* name, float* d, int wgs_per_cu) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NEXP, NFMA, NMFMA>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NEXP, NFMA, NMFMA>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per wave per inner step (r): NEXP exps, NFMA fmas, NMFMA mfmas; waves per SIMD = wgs_per_cu
    const double steps = (double)iters * 4;
    printf("%-28s wg/cu=%d  %8.3f ms  -> %7.1f ns per step per SIMD-wave-set (%d waves/SIMD)\n", name, wgs_per_cu, ms, ms * 1e6 / steps, wgs_per_cu);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3}) {
        if (w == 1) { run<8, 0, 0>("exp8", d, w); run<0, 32, 0>("fma32", d, w); run<8, 32, 0>("exp8+fma32", d, w); run<0, 0, 4>("mfma4", d, w); run<8, 0, 4>("exp8+mfma4", d, w); run<0, 32, 4>("fma32+mfma4", d, w); run<8, 32, 4>("exp8+fma32+mfma4", d, w); }
        if (w == 2) { run<8, 0, 0>("exp8", d, w); run<0, 32, 0>("fma32", d, w); run<8, 32, 0>("exp8+fma32", d, w); run<0, 0, 4>("mfma4", d, w); run<8, 0, 4>("exp8+mfma4", d, w); run<0, 32, 4>("fma32+mfma4", d, w); run<8, 32, 4>("exp8+fma32+mfma4", d, w); }
        if (w == 3) { run<8, 0, 0>("exp8", d, w); run<0, 32, 0>("fma32", d, w); run<8, 32, 0>("exp8+fma32", d, w); run<0, 0, 4>("mfma4", d, w); run<8, 0, 4>("exp8+mfma4", d, w); run<0, 32, 4>("fma32+mfma4", d, w); run<8, 32, 4>("exp8+fma32+mfma4", d, w); }
    }
    return 0;
}
