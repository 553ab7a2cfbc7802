#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float wave_sum(float x) {
    int v = __float_as_int(x);
#define DSIM_DPP_ADD(ctrl) v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false)))
    DSIM_DPP_ADD(0x111); DSIM_DPP_ADD(0x112); DSIM_DPP_ADD(0x114); DSIM_DPP_ADD(0x118);
#undef DSIM_DPP_ADD
    return (__int_as_float(__builtin_amdgcn_readlane(v, 15)) + __int_as_float(__builtin_amdgcn_readlane(v, 31))) +
           (__int_as_float(__builtin_amdgcn_readlane(v, 47)) + __int_as_float(__builtin_amdgcn_readlane(v, 63)));
}
__device__ __forceinline__ float wave_sum2(float x) {
    int v = __float_as_int(x);
#define DSIM_DPP_ADD(ctrl) v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true)))
    DSIM_DPP_ADD(0x111); DSIM_DPP_ADD(0x112); DSIM_DPP_ADD(0x114); DSIM_DPP_ADD(0x118);
#undef DSIM_DPP_ADD
    return (__int_as_float(__builtin_amdgcn_readlane(v, 15)) + __int_as_float(__builtin_amdgcn_readlane(v, 31))) +
           (__int_as_float(__builtin_amdgcn_readlane(v, 47)) + __int_as_float(__builtin_amdgcn_readlane(v, 63)));
}
__global__ void k(float* out, float* lanes) {
    float x = (float)threadIdx.x;
    out[0] = wave_sum(x);
    out[1] = wave_sum2(x);
    int v = __float_as_int(x);
    v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false)));
    lanes[threadIdx.x] = __int_as_float(v);
    int w = __float_as_int(x);
    w = __float_as_int(__int_as_float(w) + __int_as_float(__builtin_amdgcn_update_dpp(0, w, 0x118, 0xf, 0xf, false)));
    lanes[64 + threadIdx.x] = __int_as_float(w);
}
int main() {
    float *o, *l; hipMalloc(&o, 16); hipMalloc(&l, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, l);
    float h[2], hl[128]; hipMemcpy(h, o, 8, hipMemcpyDeviceToHost); hipMemcpy(hl, l, 512, hipMemcpyDeviceToHost);
    printf("sum %f %f (expect 2016)\n", h[0], h[1]);
    for (int i = 0; i < 32; ++i) printf("%g ", hl[i]); printf("\n");
    for (int i = 0; i < 32; ++i) printf("%g ", hl[64 + i]); printf("\n");
    return 0;
}
