// Development probe: what does v_permlane32_swap return for (old = x, src = x)?  Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("r0:"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\nr1:"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]); printf("\n");
    return 0;
}
