#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out, int soff, int voff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1024, 0x00020000);
    unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, voff + threadIdx.x * 4, soff, 0);
    out[threadIdx.x] = v;
}
int main() {
    unsigned *src, *out;
    hipMalloc(&src, 1 << 20); hipMalloc(&out, 256);
    unsigned h[1 << 18];
    for (int i = 0; i < (1 << 18); ++i) h[i] = 0xAB000000u + i;
    hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice);
    const int cases[][2] = {{0, 0}, {0, 2048}, {2048, 0}, {1000, 0}, {0, 1000}, {512, 512}, {1020, 0}, {0, 1020}, {1024, 0}, {0, 1024}};
    for (auto& c : cases) {
        k<<<1, 8>>>(src, out, c[0], c[1]);
        unsigned r[8]; hipMemcpy(r, out, 32, hipMemcpyDeviceToHost);
        printf("soff %5d voff %5d ->", c[0], c[1]);
        for (int i = 0; i < 8; ++i) printf(" %08x", r[i]);
        printf("\n");
    }
    return 0;
}
