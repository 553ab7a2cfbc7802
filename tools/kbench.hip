// Kernel micro-benchmark (development aid, not part of the product): times the GEMM / conv /
// attention / norm kernels on the exact shapes of one SD1.5 step (B2 U-Net batch elements) with HIP
// events, random h16 data.  Build: python tools/build_kbench.py ; run on the GPU box:
//   ./tools/kbench [B2=64] [iters=10] [filter]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../diffsim_amd/csrc/common.h"

using namespace dsim;

#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(h16* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (h16)(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = ((float)(x & 0xffff) / 32768.0f - 1.0f) * scale;
}

__global__ void maxdiff_kernel(const __bf16* a, const __bf16* b, size_t n, float* out) {
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf((float)a[i] - (float)b[i]));
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}
// bitwise difference count of two buffers (tile-order experiments must not change a single output bit)
// order-independent checksum of a 16-bit buffer (outputs of two builds of the library must agree bit for bit)
__global__ void checksum_kernel(const unsigned short* a, size_t n, unsigned long long* out) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        c += (unsigned long long)a[i] * (unsigned long long)((i * 2654435761ull) | 1ull);
    atomicAdd(out, c);
}
__global__ void bitdiff_kernel(const unsigned short* a, const unsigned short* b, size_t n, unsigned* out) {
    unsigned c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}
static void* dalloc_bf16(size_t n, unsigned seed, float scale = 1.0f) {
    void* p;
    HC(hipMalloc(&p, n * 2 + 256));
    hipLaunchKernelGGL(fill_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (h16*)p, n, seed, scale);
    return p;
}
static float* dalloc_f32(size_t n, unsigned seed, float scale = 1.0f) {
    float* p;
    HC(hipMalloc((void**)&p, n * 4 + 256));
    hipLaunchKernelGGL(fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed, scale);
    return p;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { HC(hipEventCreate(&a)); HC(hipEventCreate(&b)); }
    template <typename F> float run(F f, int iters) {
        f();
        HC(hipDeviceSynchronize());
        HC(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) f();
        HC(hipEventRecord(b, 0));
        HC(hipEventSynchronize(b));
        float ms;
        HC(hipEventElapsedTime(&ms, a, b));
        return ms / iters;
    }
};

static const char* g_filter = nullptr;
static bool want(const std::string& n) { return !g_filter || n.find(g_filter) != std::string::npos; }

static void bench_gemm(const char* name, int mode, int M, int N, int Cin, int H, int W, int epi, int iters, Timer& t,
                       void* zp, int C1 = 0, int act = 0, bool gated = false) {
    if (!want(name)) return;
    GemmArgs g;
    const int K = mode == GEMM_CONV3 ? 9 * Cin : Cin + C1;
    const int outc = epi == EPI_GEGLU ? N / 2 : N;
    void* A = dalloc_bf16((size_t)M * Cin, 1);
    void* A1 = C1 ? dalloc_bf16((size_t)M * C1, 5) : nullptr;
    void* Wt = dalloc_bf16((size_t)N * K, 2, 0.05f);
    float* bias = dalloc_f32(N, 3);
    void* res = dalloc_bf16((size_t)M * outc, 4);
    void* out;
    HC(hipMalloc(&out, (size_t)M * outc * 2));
    g.A0 = A; g.C0 = Cin; g.A1 = A1; g.C1 = C1; g.mode = mode; g.Hin = g.Hout = H; g.Win = g.Wout = W; g.M = M; g.N = N; g.K = K;
    g.W = Wt; g.bias = bias; g.epi = epi; g.residual = epi == EPI_RESIDUAL ? res : nullptr; g.out = out; g.ldo = outc;
    g.zero_page = zp;
    g.act = act;
    if (gated) { g.gate = dalloc_f32(N, 7); g.rows_per_batch = 256; }
    if (epi == EPI_GEGLU) g.geglu_blk = geglu_block_rows(N);      // KB_GEXP=8192: the 32-row blocks (256 / 128-column tiles) beside it
    int st = DSIM_OK;
    const double fl = 2.0 * M * (double)N * K;
    float msv[3];
    const int forces[3] = {128, 256, 0};
    for (int v = 0; v < 3; ++v) {
        g_force_bm = forces[v];
        g_gemm_persistent = 1;
        if (v == 0 && getenv("KB_NOPERSIST")) { g_force_bm = 0; g_gemm_persistent = 0; }   // column 1 = auto tiles, one tile per workgroup
        msv[v] = t.run([&] { st = launch_gemm(g, DSIM_BF16, 0); }, iters);
    }
    if (getenv("KB_SKINNY") && atoi(getenv("KB_SKINNY")) == 2) g_gemm_skinny = 2;      // widened applies() rule for the sweep
    if (getenv("KB_SKINNY") && gemm_skinny_applies(g)) {      // small-batch kernel: gemm_kernel against every skinny tile (interleaved rounds)
        const int tiles[8] = {-1, (64 << 8) | 64, (128 << 8) | 64, (64 << 8) | 128, (128 << 8) | 128, (128 << 8) | 160, (64 << 8) | 80, (128 << 8) | 80};
        const int rounds = getenv("KB_ROUNDS") ? atoi(getenv("KB_ROUNDS")) : 5;
        std::vector<std::vector<float>> ms(8);
        g_force_bm = 0;
        for (int r = 0; r < rounds; ++r)
            for (int k = 0; k < 8; ++k) {
                g_gemm_skinny = tiles[k] >= 0 ? (atoi(getenv("KB_SKINNY")) == 2 ? 2 : 1) : 0; g_skinny_tile = tiles[k] > 0 ? tiles[k] : 0;
                ms[k].push_back(t.run([&] { st = launch_gemm(g, DSIM_BF16, 0); }, iters));
            }
        g_gemm_skinny = 1; g_skinny_tile = 0;
        int hb, hn;
        gemm_skinny_tile(g, &hb, &hn);
        printf("  small-batch median ms: gemm_kernel %.4f | 64x64 %.4f | 128x64 %.4f | 64x128 %.4f | 128x128 %.4f | 128x160 %.4f | 64x80 %.4f | 128x80 %.4f | heuristic %dx%d\n",
               (std::sort(ms[0].begin(), ms[0].end()), ms[0][rounds / 2]), (std::sort(ms[1].begin(), ms[1].end()), ms[1][rounds / 2]),
               (std::sort(ms[2].begin(), ms[2].end()), ms[2][rounds / 2]), (std::sort(ms[3].begin(), ms[3].end()), ms[3][rounds / 2]),
               (std::sort(ms[4].begin(), ms[4].end()), ms[4][rounds / 2]), (std::sort(ms[5].begin(), ms[5].end()), ms[5][rounds / 2]),
               (std::sort(ms[6].begin(), ms[6].end()), ms[6][rounds / 2]), (std::sort(ms[7].begin(), ms[7].end()), ms[7][rounds / 2]), hb, hn);
    }
    if (const char* e = getenv("KB_GEXP")) {            // kernel experiment masks on the auto tile, e.g. KB_GEXP=32,64,128
        // interleaved rounds in one process (cdna_hip_programming.md rule 24): every mask (0 first) timed once per round, min and median
        std::vector<int> masks{0};
        std::string l = e;
        for (size_t pos = 0; pos < l.size();) {
            size_t nx = l.find(',', pos);
            if (nx == std::string::npos) nx = l.size();
            masks.push_back(atoi(l.substr(pos, nx - pos).c_str()));
            pos = nx + 1;
        }
        const int rounds = getenv("KB_ROUNDS") ? atoi(getenv("KB_ROUNDS")) : 5;
        std::vector<std::vector<float>> ms(masks.size());
        for (int r = 0; r < rounds; ++r)
            for (size_t k = 0; k < masks.size(); ++k) {
                g_gemm_exp = masks[k];
                ms[k].push_back(t.run([&] { st = launch_gemm(g, DSIM_BF16, 0); }, iters));
            }
        // outputs of every mask against mask 0, bit for bit
        void* ref;
        unsigned* dcount;
        HC(hipMalloc(&ref, (size_t)M * outc * 2));
        HC(hipMalloc((void**)&dcount, 4));
        g_gemm_exp = 0;
        st = launch_gemm(g, DSIM_BF16, 0);
        HC(hipMemcpy(ref, out, (size_t)M * outc * 2, hipMemcpyDeviceToDevice));
        std::vector<unsigned> nd(masks.size(), 0);
        for (size_t k = 1; k < masks.size(); ++k) {
            g_gemm_exp = masks[k];
            HC(hipMemset(out, 0xff, (size_t)M * outc * 2));
            st = launch_gemm(g, DSIM_BF16, 0);
            HC(hipMemset(dcount, 0, 4));
            hipLaunchKernelGGL(bitdiff_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned short*)out, (const unsigned short*)ref, (size_t)M * outc, dcount);
            HC(hipMemcpy(&nd[k], dcount, 4, hipMemcpyDeviceToHost));
        }
        HC(hipFree(ref)); HC(hipFree(dcount));
        g_gemm_exp = 0;
        printf("  gemm exp min/median ms [differing outputs]:");
        for (size_t k = 0; k < masks.size(); ++k) {
            std::sort(ms[k].begin(), ms[k].end());
            printf("  %d:%.3f/%.3f [%u]", masks[k], ms[k][0], ms[k][rounds / 2], nd[k]);
        }
        printf("\n");
    }
#ifdef DSIM_STAMPS
    {   // phase stamps of gemm_kernel on the auto tile: share of a wave's cycles per phase
        unsigned long long* sb;
        HC(hipMalloc((void**)&sb, 64));
        HC(hipMemset(sb, 0, 64));
        g_force_bm = 0; g_gemm_stamps = sb;
        st = launch_gemm(g, DSIM_BF16, 0);
        HC(hipDeviceSynchronize());
        g_gemm_stamps = nullptr;
        unsigned long long h[8];
        HC(hipMemcpy(h, sb, 64, hipMemcpyDeviceToHost));
        double tot = 0;
        for (int i = 0; i < 6; ++i) tot += (double)h[i];
        static const char* nm[6] = {"init", "top-wait", "K-loop", "setup+stage0", "epilogue", "re-derive"};
        printf("  stamps (%% of wave cycles, %llu workgroups):", h[6]);
        for (int i = 0; i < 6; ++i) printf("  %s %.1f", nm[i], 100.0 * (double)h[i] / tot);
        printf("\n");
        HC(hipFree(sb));
    }
#endif
    int bm, bn;
    gemm_launch_tile(g, DSIM_BF16, &bm, &bn);
    unsigned long long csum = 0;
    {
        unsigned long long* dc;
        HC(hipMalloc((void**)&dc, 8));
        HC(hipMemset(dc, 0, 8));
        g_force_bm = 0;
        HC(hipMemset(out, 0xff, (size_t)M * outc * 2));
        st = launch_gemm(g, DSIM_BF16, 0);
        hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned short*)out, (size_t)M * outc, dc);
        HC(hipMemcpy(&csum, dc, 8, hipMemcpyDeviceToHost));
        HC(hipFree(dc));
    }
    printf("%-26s M=%7d N=%5d K=%5d  bm128 %7.3f ms %6.1f TF | bm256 %7.3f ms %6.1f TF | auto %dx%d %7.3f ms %6.1f TF st=%d sum=%016llx\n",
           name, M, N, K, msv[0], fl / msv[0] / 1e9, msv[1], fl / msv[1] / 1e9, bm, bn, msv[2], fl / msv[2] / 1e9, st, csum);
    HC(hipFree(A)); if (A1) HC(hipFree(A1)); HC(hipFree(Wt)); HC(hipFree(bias)); HC(hipFree(res)); HC(hipFree(out));
}

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float r; memcpy(&r, &u, 4); return r; }

static void bench_attn(const char* name, int B, int Bkv, int H, int Nq, int Nk, int D, int iters, Timer& t) {
    if (!want(name)) return;
    const int C = H * D;
    const bool self = Bkv == B && Nq == Nk;
    void* q = dalloc_bf16((size_t)B * Nq * (self ? 3 * C : C), 1);
    void* kv = self ? nullptr : dalloc_bf16((size_t)Bkv * Nk * 2 * C, 2);
    void* out;
    HC(hipMalloc(&out, (size_t)B * Nq * C * 2));
    AttnArgs a;
    if (self) { a.q = q; a.ldq = 3 * C; a.k = (char*)q + C * 2; a.v = (char*)q + 2 * C * 2; a.ldk = 3 * C; }
    else { a.q = q; a.ldq = C; a.k = kv; a.v = (char*)kv + C * 2; a.ldk = 2 * C; }
    a.out = out; a.ldo = C; a.B = B; a.Bkv = Bkv; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D;
    int st = DSIM_OK;
    const int rounds = getenv("KB_ROUNDS") ? atoi(getenv("KB_ROUNDS")) : 5;
    std::vector<float> m0, m1;
    for (int r = 0; r < rounds; ++r) {            // interleaved rounds: plain block order / XCD-aware order
        a.xcd_remap = 0;
        m0.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
        a.xcd_remap = 1;
        m1.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
    }
    if (Nk <= 96) {                               // keys resident in LDS (attn_short_kernel; round 5's form of it) against the tiled kernel
        std::vector<float> q1, q2, q3;
        const size_t no = (size_t)B * Nq * C;
        std::vector<unsigned short> h0(no), h1(no);
        g_attn_short = 0;
        HC(hipMemset(out, 0xff, no * 2));
        st = launch_attention(a, DSIM_BF16, 0);
        HC(hipMemcpy(h0.data(), out, no * 2, hipMemcpyDeviceToHost));
        g_attn_short = 1;
        HC(hipMemset(out, 0xff, no * 2));
        st = launch_attention(a, DSIM_BF16, 0);
        HC(hipMemcpy(h1.data(), out, no * 2, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; size_t nbad = 0;
        for (size_t i = 0; i < no; ++i) {
            const float x = bf2f(h0[i]), y = bf2f(h1[i]);
            if (!(y - y == 0.0f)) ++nbad;
            md = std::max(md, (double)fabsf(x - y)); mx = std::max(mx, (double)fabsf(x));
        }
        for (int r = 0; r < rounds; ++r) {
            g_attn_short = 0;
            q1.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
            g_attn_short = 2;
            q3.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
            g_attn_short = 1;
            q2.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
        }
        if (getenv("KB_SHORTABL")) {              // ablations of the short-key kernel: 1 no output stores, 2 no Q prefetch, 3 neither, 4 the output as coalesced 1 KB stores
            for (int m = 1; m <= 4; ++m) {
                g_attn_dbg = m; g_attn_short = 1;
                std::vector<float> qa;
                for (int r = 0; r < rounds; ++r) qa.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
                std::sort(qa.begin(), qa.end());
                printf("  short-key kernel ablation %d: %8.3f ms\n", m, qa[rounds / 2]);
            }
            g_attn_dbg = 0;
        }
        std::sort(q1.begin(), q1.end()); std::sort(q2.begin(), q2.end()); std::sort(q3.begin(), q3.end());
        printf("  tiled kernel %8.3f/%8.3f ms | short-key kernel, round 5 %8.3f/%8.3f ms | short-key kernel %8.3f/%8.3f ms (min/median)   vs tiled: max |diff| %.3g of max %.3g, %zu non-finite\n",
               q1[0], q1[rounds / 2], q3[0], q3[rounds / 2], q2[0], q2[rounds / 2], md, mx, nbad);
    }
    if (self && D == 160 && Nq == 256) {          // the tiled kernel against the persistent core (attn160.hip), and their outputs against each other
        std::vector<float> q1, q2;
        const size_t no = (size_t)B * Nq * C;
        std::vector<unsigned short> h0(no), h1(no);
        g_sdpa160 = 0;
        HC(hipMemset(out, 0xff, no * 2));
        st = launch_attention(a, DSIM_BF16, 0);
        HC(hipMemcpy(h0.data(), out, no * 2, hipMemcpyDeviceToHost));
        g_sdpa160 = 1;
        HC(hipMemset(out, 0xff, no * 2));
        st = launch_attention(a, DSIM_BF16, 0);
        HC(hipMemcpy(h1.data(), out, no * 2, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; size_t nbad = 0, ndiff = 0;
        for (size_t i = 0; i < no; ++i) {
            const float x = bf2f(h0[i]), y = bf2f(h1[i]);
            if (!(y - y == 0.0f)) ++nbad;
            if (h0[i] != h1[i]) ++ndiff;
            md = std::max(md, (double)fabsf(x - y)); mx = std::max(mx, (double)fabsf(x));
        }
        for (int r = 0; r < rounds; ++r) {
            g_sdpa160 = 0;
            q1.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
            g_sdpa160 = 1;
            q2.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
        }
        std::sort(q1.begin(), q1.end()); std::sort(q2.begin(), q2.end());
        printf("  tiled kernel %8.4f/%8.4f ms | persistent core %8.4f/%8.4f ms (min/median)   outputs: max |diff| %.3g of max %.3g, %zu of %zu differ, %zu non-finite\n",
               q1[0], q1[rounds / 2], q2[0], q2[rounds / 2], md, mx, ndiff, no, nbad);
    }
    if (Nk >= 256 && Nk < 2048) {                 // exact running maximum against the fixed-reference softmax at mid-length key sequences
        std::vector<float> q1, q2;
        for (int r = 0; r < rounds; ++r) {
            g_attn_fast_min = 2048;
            q1.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
            g_attn_fast_min = 256;
            q2.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
        }
        g_attn_fast_min = 1024;
        std::sort(q1.begin(), q1.end()); std::sort(q2.begin(), q2.end());
        printf("  running maximum %8.3f/%8.3f ms | fixed reference %8.3f/%8.3f ms (min/median)\n", q1[0], q1[rounds / 2], q2[0], q2[rounds / 2]);
    }
    if (Nk >= 2048 || (D == 64 && Nk > 96)) {   // one query block per wave (attn_kernel) against two (attn_long_kernel)
        std::vector<float> q1, q2;
        for (int r = 0; r < rounds; ++r) {
            g_attn_q2 = 0;
            q1.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
            g_attn_q2 = 1;
            q2.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
        }
        std::sort(q1.begin(), q1.end()); std::sort(q2.begin(), q2.end());
        if (const char* e = getenv("KB_ATDBG")) {
            std::vector<int> vals{0};
            std::string l = e;
            for (size_t pos = 0; pos < l.size();) {
                size_t nx = l.find(',', pos);
                if (nx == std::string::npos) nx = l.size();
                vals.push_back(atoi(l.substr(pos, nx - pos).c_str()));
                pos = nx + 1;
            }
            std::vector<std::vector<float>> ms(vals.size());
            for (int r = 0; r < rounds; ++r)
                for (size_t k = 0; k < vals.size(); ++k) {
                    g_attn_dbg = vals[k];
                    ms[k].push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
                }
            g_attn_dbg = 0;
            printf("  attn_long ablation median ms:");
            for (size_t k = 0; k < vals.size(); ++k) {
                std::sort(ms[k].begin(), ms[k].end());
                printf("  %d:%.3f", vals[k], ms[k][rounds / 2]);
            }
            printf("\n");
        }
        printf("  32 queries per wave %8.3f/%8.3f ms | 64 queries per wave %8.3f/%8.3f ms (min/median)\n", q1[0], q1[rounds / 2], q2[0], q2[rounds / 2]);
    }
    if (const char* e = getenv("KB_ATTNPAD")) {          // occupancy probe: KB of unused LDS per workgroup (tiled kernels)
        std::string l = e;
        for (size_t pos = 0; pos < l.size();) {
            size_t nx = l.find(',', pos);
            if (nx == std::string::npos) nx = l.size();
            g_attn_lds_pad = atoi(l.substr(pos, nx - pos).c_str());
            std::vector<float> qa;
            for (int r = 0; r < rounds; ++r) qa.push_back(t.run([&] { st = launch_attention(a, DSIM_BF16, 0); }, iters));
            std::sort(qa.begin(), qa.end());
            printf("  +%d KB of LDS per workgroup: %8.3f ms (st=%d)\n", g_attn_lds_pad, qa[rounds / 2], st);
            pos = nx + 1;
        }
        g_attn_lds_pad = 0;
    }
    std::sort(m0.begin(), m0.end()); std::sort(m1.begin(), m1.end());
    const float ms = m1[rounds / 2];
    const double fl = 4.0 * B * H * (double)Nq * Nk * D;
    printf("%-28s B=%3d H=%d Nq=%5d Nk=%5d D=%3d  plain %8.3f/%8.3f ms | xcd-aware min/median %8.3f/%8.3f ms  %7.1f TF/s  st=%d\n", name, B, H,
           Nq, Nk, D, m0[0], m0[rounds / 2], m1[0], ms, fl / ms / 1e9, st);
    HC(hipFree(q)); if (kv) HC(hipFree(kv)); HC(hipFree(out));
}



// host reference of one pair's score (double precision; the SDPA outputs rounded to bf16 as the 16-bit kernels do)
static float bf16_round(float x) { unsigned u; memcpy(&u, &x, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; float r; memcpy(&r, &u, 4); return r; }
static double tail_reference(const std::vector<unsigned short>& q, const std::vector<unsigned short>& k, const std::vector<unsigned short>& v,
                             int ia, int ib, int B, int H, int N, int D, int mse) {
    auto f = [](unsigned short h) { unsigned u = (unsigned)h << 16; float r; memcpy(&r, &u, 4); return (double)r; };
    const int C = H * D;
    const size_t img = (size_t)B * N * C;
    double res = 0;
    std::vector<double> os((size_t)N * D), ox((size_t)N * D), p(N);
    for (int dir = 0; dir < 2; ++dir) {
        const int iq = dir ? ib : ia, ix = dir ? ia : ib;
        double sxy = 0, sxx = 0, syy = 0, sd = 0;
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h) {
                const size_t off = (size_t)b * N * C + h * D;
                for (int pass = 0; pass < 2; ++pass) {
                    const int ik = pass ? ix : iq;
                    std::vector<double>& o = pass ? ox : os;
                    for (int i = 0; i < N; ++i) {
                        double mx = -1e300;
                        for (int j = 0; j < N; ++j) {
                            double a = 0;
                            for (int d = 0; d < D; ++d) a += f(q[iq * img + off + (size_t)i * C + d]) * f(k[ik * img + off + (size_t)j * C + d]);
                            p[j] = a / sqrt((double)D);
                            mx = std::max(mx, p[j]);
                        }
                        double l = 0;
                        for (int j = 0; j < N; ++j) { p[j] = exp(p[j] - mx); l += p[j]; }
                        for (int d = 0; d < D; ++d) {
                            double a = 0;
                            for (int j = 0; j < N; ++j) a += p[j] * f(v[ik * img + off + (size_t)j * C + d]);
                            o[(size_t)i * D + d] = (double)bf16_round((float)(a / l));
                        }
                    }
                }
                for (size_t e = 0; e < (size_t)N * D; ++e) {
                    sxy += ox[e] * os[e]; sxx += ox[e] * ox[e]; syy += os[e] * os[e]; sd += (ox[e] - os[e]) * (ox[e] - os[e]);
                }
            }
        res += mse ? sd / ((double)B * H * N * D) : sxy / (std::max(sqrt(sxx), 1e-8) * std::max(sqrt(syy), 1e-8));
    }
    return res * 0.5;
}

// the fused score tail at SD1.5's default tap (256 tokens x 8 heads x 160): pair_tail_kernel against pair_tail160_kernel (attn160.hip)
static void bench_tail(const char* name, int np, int iters, Timer& t) {
    if (!want(name)) return;
    const int B = 2, H = 8, N = 256, D = 160, C = H * D;
    const size_t per = (size_t)2 * np * B * N * C;
    void* q = dalloc_bf16(per, 11, 2.0f);
    void* k = dalloc_bf16(per, 12, 2.0f);
    void* v = dalloc_bf16(per, 13, 1.0f);
    std::vector<int32_t> ha(np), hb(np);
    for (int i = 0; i < np; ++i) { ha[i] = 2 * i; hb[i] = getenv("KB_TAILSELF") ? 2 * i : 2 * i + 1; }
    int32_t *ia, *ib;
    HC(hipMalloc((void**)&ia, np * 4)); HC(hipMalloc((void**)&ib, np * 4));
    HC(hipMemcpy(ia, ha.data(), np * 4, hipMemcpyHostToDevice)); HC(hipMemcpy(ib, hb.data(), np * 4, hipMemcpyHostToDevice));
    const size_t sb = pair_score_scratch_bytes(np, B, H, N, D);
    void* scratch; HC(hipMalloc(&scratch, sb));
    float *o0, *o1;
    HC(hipMalloc((void**)&o0, np * 4)); HC(hipMalloc((void**)&o1, np * 4));
    int st0 = 0, st1 = 0;
    const int rounds = getenv("KB_ROUNDS") ? atoi(getenv("KB_ROUNDS")) : 5;
    for (int sim = 0; sim < 2; ++sim) {
        std::vector<float> m0, m1;
        for (int r = 0; r < rounds; ++r) {
            g_tail160 = 0;
            m0.push_back(t.run([&] { st0 = launch_pair_score(q, k, v, ia, ib, np, B, H, N, D, DSIM_BF16, sim, o0, scratch, sb, 0); }, iters));
            g_tail160 = 1;
            m1.push_back(t.run([&] { st1 = launch_pair_score(q, k, v, ia, ib, np, B, H, N, D, DSIM_BF16, sim, o1, scratch, sb, 0); }, iters));
        }
        HC(hipDeviceSynchronize());
        std::vector<float> h0(np), h1(np);
        HC(hipMemcpy(h0.data(), o0, np * 4, hipMemcpyDeviceToHost)); HC(hipMemcpy(h1.data(), o1, np * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (int i = 0; i < np; ++i) md = std::max(md, (double)fabsf(h0[i] - h1[i]) / std::max(1e-12, (double)fabsf(h0[i])));
        std::sort(m0.begin(), m0.end()); std::sort(m1.begin(), m1.end());
        const double fl = 2.0 * np * B * H * 2 * 4.0 * N * N * D;
        printf("%-20s %s pairs=%3d  tiled %8.4f/%8.4f ms %7.1f TF/s | persistent d160 %8.4f/%8.4f ms %7.1f TF/s (min/median)  max rel diff %.3g  score[0] %.6f / %.6f st=%d/%d\n",
               name, sim ? "mse   " : "cosine", np, m0[0], m0[rounds / 2], fl / m0[rounds / 2] / 1e9, m1[0], m1[rounds / 2], fl / m1[rounds / 2] / 1e9, md,
               h0[0], h1[0], st0, st1);
#ifdef DSIM_STAMPS
        if (sim == 0) {
            const int nwg = 256;
            unsigned long long* sbuf; HC(hipMalloc((void**)&sbuf, (size_t)nwg * 8 * 8 * 8)); HC(hipMemset(sbuf, 0, (size_t)nwg * 8 * 8 * 8));
            g_tail160_dbg = (float*)sbuf; g_tail160 = 1;
            launch_pair_score(q, k, v, ia, ib, np, B, H, N, D, DSIM_BF16, sim, o1, scratch, sb, 0);
            HC(hipDeviceSynchronize());
            g_tail160_dbg = nullptr;
            std::vector<unsigned long long> hs((size_t)nwg * 64);
            HC(hipMemcpy(hs.data(), sbuf, hs.size() * 8, hipMemcpyDeviceToHost));
            const char* nm[8] = {"QK", "yload+Vpre+softmax", "PV part 1", "lgkm+vmcnt wait", "barrier", "DMA issue+K pre", "PV part 2", "epilogues"};
            for (int cls = 0; cls < 2; ++cls) {
                double tot[8] = {0}, all = 0; int n = 0;
                for (int w = 0; w < nwg * 8; ++w) { if (((w & 7) < 4) != (cls == 0)) continue; if (!hs[(size_t)w * 8]) continue; ++n; for (int i = 0; i < 8; ++i) { tot[i] += hs[(size_t)w * 8 + i]; all += hs[(size_t)w * 8 + i]; } }
                printf("   stamps, waves %s (%d waves, %.0f cycles per wave):", cls ? "4-7" : "0-3", n, n ? all / n : 0.0);
                for (int i = 0; i < 8; ++i) printf("  %s %.1f%%", nm[i], all ? 100.0 * tot[i] / all : 0.0);
                printf("\n");
            }
            HC(hipFree(sbuf));
        }
#endif
        if (getenv("KB_TAILDBG") && sim == 0) {
            // the first unit (pair 0, direction 0, b 0, head 0): both attention outputs against a host evaluation, error by key-independent position
            float* dbg; HC(hipMalloc((void**)&dbg, 2 * N * D * 4)); HC(hipMemset(dbg, 0, 2 * N * D * 4));
            g_tail160_dbg = dbg; g_tail160 = 1;
            launch_pair_score(q, k, v, ia, ib, np, B, H, N, D, DSIM_BF16, sim, o1, scratch, sb, 0);
            HC(hipDeviceSynchronize());
            g_tail160_dbg = nullptr;
            std::vector<float> hd(2 * N * D);
            HC(hipMemcpy(hd.data(), dbg, hd.size() * 4, hipMemcpyDeviceToHost));
            std::vector<unsigned short> hq(per), hk(per), hv(per);
            HC(hipMemcpy(hq.data(), q, per * 2, hipMemcpyDeviceToHost)); HC(hipMemcpy(hk.data(), k, per * 2, hipMemcpyDeviceToHost));
            HC(hipMemcpy(hv.data(), v, per * 2, hipMemcpyDeviceToHost));
            auto f = [](unsigned short h) { unsigned u = (unsigned)h << 16; float r; memcpy(&r, &u, 4); return (double)r; };
            const size_t img = (size_t)B * N * C;
            std::vector<double> p2v(N);
            for (int pass = 0; pass < 2; ++pass) {
                const int iq = ha[0], ik = pass ? hb[0] : ha[0];
                double worst = 0; int wi = -1, wd = -1;
                std::vector<double> errq(N, 0.0), errd(D, 0.0), p(N);
                for (int i = 0; i < N; ++i) {
                    double mx = -1e300;
                    for (int j = 0; j < N; ++j) {
                        double a = 0;
                        for (int d = 0; d < D; ++d) a += f(hq[iq * img + (size_t)i * C + d]) * f(hk[ik * img + (size_t)j * C + d]);
                        p[j] = a / sqrt((double)D); mx = std::max(mx, p[j]);
                    }
                    double l = 0;
                    for (int j = 0; j < N; ++j) { p[j] = exp(p[j] - mx); l += p[j]; }
                    for (int d = 0; d < D; ++d) {
                        double a = 0;
                        for (int j = 0; j < N; ++j) a += p[j] * f(hv[ik * img + (size_t)j * C + d]);
                        const double e = fabs(a / l - hd[((size_t)pass * N + i) * D + d]);
                        errq[i] = std::max(errq[i], e); errd[d] = std::max(errd[d], e);
                        if (e > worst) { worst = e; wi = i; wd = d; }
                    }
                }
                printf("   dbg pass %d: worst |err| %.4g at query %d d %d\n     per query:", pass, worst, wi, wd);
                for (int i = 0; i < N; ++i) printf(" %.2g", errq[i]);
                printf("\n     per d:");
                for (int d = 0; d < D; ++d) printf(" %.2g", errd[d]);
                printf("\n");
            }
            {
                if (getenv("KB_TAILSELF")) {
                    double md = 0; int cnt = 0, wq = -1, wd = -1;
                    for (int e = 0; e < N * D; ++e) { const double dd = fabs((double)hd[e] - (double)hd[(size_t)N * D + e]); if (dd > 0) ++cnt; if (dd > md) { md = dd; wq = e / D; wd = e % D; } }
                    printf("   self pair: pass 0 vs pass 1 outputs differ in %d of %d elements, max |diff| %.3g at query %d d %d\n", cnt, N * D, md, wq, wd);
                    for (int w = 0; w < 8; ++w) { int c2 = 0; for (int e = w * 32 * D; e < (w + 1) * 32 * D; ++e) c2 += hd[e] != hd[(size_t)N * D + e]; printf(" wave %d: %d", w, c2); }
                    printf("\n");
                }
                printf("   sums of the dumped f32 outputs, unit 0 per wave [x.y x.x y.y] (f32 values and rounded to bf16):\n    ");
                for (int w = 0; w < 8; ++w) {
                    double a = 0, b2 = 0, c2 = 0, ar = 0, br = 0, cr = 0;
                    for (int e = w * 32 * D; e < (w + 1) * 32 * D; ++e) {
                        const double y = hd[e], x = hd[(size_t)N * D + e], yr = bf16_round(hd[e]), xr = bf16_round(hd[(size_t)N * D + e]);
                        a += x * y; b2 += x * x; c2 += y * y; ar += xr * yr; br += xr * xr; cr += yr * yr;
                    }
                    printf(" [%.4f %.3f %.3f | %.4f %.3f %.3f]", a, b2, c2, ar, br, cr);
                }
                printf("\n");
            }
            HC(hipFree(dbg));
            // per-unit partial sums of pair 0 against the host (x.y, x.x, y.y)
            std::vector<float> hp((size_t)2 * B * H * 8 * 4);
            HC(hipMemcpy(hp.data(), scratch, hp.size() * 4, hipMemcpyDeviceToHost));
            std::vector<double> os((size_t)N * D), ox((size_t)N * D);
            for (int dir = 0; dir < 2; ++dir)
                for (int b = 0; b < B; ++b)
                    for (int h = 0; h < H; ++h) {
                        const int iq = dir ? hb[0] : ha[0], ix = dir ? ha[0] : hb[0];
                        const size_t off = (size_t)b * N * C + h * D;
                        for (int pass = 0; pass < 2; ++pass) {
                            const int ik = pass ? ix : iq;
                            std::vector<double>& o = pass ? ox : os;
                            for (int i = 0; i < N; ++i) {
                                double mx = -1e300;
                                for (int j = 0; j < N; ++j) {
                                    double a = 0;
                                    for (int d = 0; d < D; ++d) a += f(hq[iq * img + off + (size_t)i * C + d]) * f(hk[ik * img + off + (size_t)j * C + d]);
                                    p2v[j] = a / sqrt((double)D); mx = std::max(mx, p2v[j]);
                                }
                                double l = 0;
                                for (int j = 0; j < N; ++j) { p2v[j] = exp(p2v[j] - mx); l += p2v[j]; }
                                for (int d = 0; d < D; ++d) {
                                    double a = 0;
                                    for (int j = 0; j < N; ++j) a += p2v[j] * f(hv[ik * img + off + (size_t)j * C + d]);
                                    o[(size_t)i * D + d] = (double)bf16_round((float)(a / l));
                                }
                            }
                        }
                        printf("   unit dir %d b %d h %d:", dir, b, h);
                        for (int w = 0; w < 8; ++w) {
                            double sxy = 0, sxx = 0, syy = 0;
                            for (size_t e = (size_t)w * 32 * D; e < (size_t)(w + 1) * 32 * D; ++e) { sxy += ox[e] * os[e]; sxx += ox[e] * ox[e]; syy += os[e] * os[e]; }
                            const float* g = &hp[(((size_t)dir * B * H + b * H + h) * 8 + w) * 4];
                            printf(" [%.4f/%.4f %.3f/%.3f %.3f/%.3f]", g[0], sxy, g[1], sxx, g[2], syy);
                        }
                        printf("\n");
                    }
        }
        if (const char* e = getenv("KB_TAILEXP")) {          // interleaved rounds over experiment masks of the persistent kernel
            std::vector<int> vals{0};
            std::string l = e;
            for (size_t pos = 0; pos < l.size();) {
                size_t nx = l.find(',', pos);
                if (nx == std::string::npos) nx = l.size();
                vals.push_back(atoi(l.substr(pos, nx - pos).c_str()));
                pos = nx + 1;
            }
            std::vector<std::vector<float>> ms(vals.size());
            std::vector<float> sc(vals.size());
            g_tail160 = 1;
            for (int r = 0; r < rounds; ++r)
                for (size_t kk = 0; kk < vals.size(); ++kk) {
                    g_tail160_exp = vals[kk];
                    ms[kk].push_back(t.run([&] { st1 = launch_pair_score(q, k, v, ia, ib, np, B, H, N, D, DSIM_BF16, sim, o1, scratch, sb, 0); }, iters));
                    HC(hipMemcpy(&sc[kk], o1 + (np - 1), 4, hipMemcpyDeviceToHost));
                }
            g_tail160_exp = 0;
            printf("   experiment masks, median ms (score of the last pair):");
            for (size_t kk = 0; kk < vals.size(); ++kk) { std::sort(ms[kk].begin(), ms[kk].end()); printf("  %d: %.4f (%.7f)", vals[kk], ms[kk][rounds / 2], sc[kk]); }
            printf("\n");
        }
        if (getenv("KB_TAILREF")) {
            std::vector<unsigned short> hq(per), hk(per), hv(per);
            HC(hipMemcpy(hq.data(), q, per * 2, hipMemcpyDeviceToHost)); HC(hipMemcpy(hk.data(), k, per * 2, hipMemcpyDeviceToHost));
            HC(hipMemcpy(hv.data(), v, per * 2, hipMemcpyDeviceToHost));
            const int pi = np - 1;
            printf("   host reference pair %d: %.7f   tiled %.7f   persistent %.7f\n", pi, tail_reference(hq, hk, hv, ha[pi], hb[pi], B, H, N, D, sim), h0[pi], h1[pi]);
        }
    }
    HC(hipFree(q)); HC(hipFree(k)); HC(hipFree(v)); HC(hipFree(ia)); HC(hipFree(ib)); HC(hipFree(scratch)); HC(hipFree(o0)); HC(hipFree(o1));
}

static void bench_gn(const char* name, int B, int HW, int C, int iters, Timer& t) {
    if (!want(name)) return;
    void* x = dalloc_bf16((size_t)B * HW * C, 1);
    float* g = dalloc_f32(C, 2);
    float* b = dalloc_f32(C, 3);
    void *out, *sc;
    HC(hipMalloc(&out, (size_t)B * HW * C * 2));
    HC(hipMalloc(&sc, groupnorm_scratch_bytes(B, 32)));
    int st = DSIM_OK;
    if (const char* e = getenv("KB_NORMPAD")) {
        std::string l = e;
        int st2 = 0;
        for (size_t pos = 0; pos < l.size();) {
            size_t nx = l.find(',', pos);
            if (nx == std::string::npos) nx = l.size();
            g_norm_lds_pad = atoi(l.substr(pos, nx - pos).c_str());
            std::vector<float> qa;
            for (int r = 0; r < 5; ++r) qa.push_back(t.run([&] { st2 = launch_groupnorm(x, C, nullptr, 0, g, b, out, B, HW, 32, 1e-5f, 1, DSIM_BF16, sc, 0); }, iters));
            std::sort(qa.begin(), qa.end());
            printf("  %s +%d KB of LDS per workgroup: %8.4f ms (st=%d)\n", name, g_norm_lds_pad, qa[2], st2);
            pos = nx + 1;
        }
        g_norm_lds_pad = 0;
    }
    const float ms = t.run([&] { st = launch_groupnorm(x, C, nullptr, 0, g, b, out, B, HW, 32, 1e-5f, 1, DSIM_BF16, sc, 0); }, iters);
    if (getenv("KB_GN_SILU")) {      // with and without the SiLU, interleaved: is the apply pass VALU-limited?
        std::vector<float> a, c;
        for (int r = 0; r < 5; ++r) {
            a.push_back(t.run([&] { st = launch_groupnorm(x, C, nullptr, 0, g, b, out, B, HW, 32, 1e-5f, 1, DSIM_BF16, sc, 0); }, iters));
            c.push_back(t.run([&] { st = launch_groupnorm(x, C, nullptr, 0, g, b, out, B, HW, 32, 1e-5f, 0, DSIM_BF16, sc, 0); }, iters));
        }
        std::sort(a.begin(), a.end()); std::sort(c.begin(), c.end());
        printf("  median ms with SiLU %.3f, without %.3f\n", a[2], c[2]);
    }
    const double by = 3.0 * B * (double)HW * C * 2;
    printf("%-28s B=%3d HW=%5d C=%5d                 %8.3f ms  %7.1f GB/s  st=%d\n", name, B, HW, C, ms, by / ms / 1e6, st);
    if (getenv("KB_GNCHUNK")) {      // statistics + apply per sub-batch of images (does the apply pass then read from the Infinity Cache?)
        const int rounds = getenv("KB_ROUNDS") ? atoi(getenv("KB_ROUNDS")) : 5;
        const int cs[5] = {B, 128, 64, 32, 16};
        std::vector<std::vector<float>> msr(5);
        for (int r = 0; r < rounds; ++r)
            for (int k = 0; k < 5; ++k) {
                const int cb = cs[k];
                msr[k].push_back(t.run([&] {
                    for (int b0 = 0; b0 < B; b0 += cb)
                        st |= launch_groupnorm((const char*)x + (size_t)b0 * HW * C * 2, C, nullptr, 0, g, b, (char*)out + (size_t)b0 * HW * C * 2,
                                               std::min(cb, B - b0), HW, 32, 1e-5f, 1, DSIM_BF16, sc, 0);
                }, iters));
            }
        printf("  images per launch pair -> median ms:");
        for (int k = 0; k < 5; ++k) { std::sort(msr[k].begin(), msr[k].end()); printf("  %d: %.3f", cs[k], msr[k][rounds / 2]); }
        printf("\n");
    }
    const float ms2 = t.run([&] { st = launch_layernorm(x, g, b, out, B * HW, C, 1e-5f, DSIM_BF16, 0); }, iters);
    printf("%-28s M=%7d C=%5d                        %8.3f ms  %7.1f GB/s  st=%d\n", (std::string(name) + "_ln").c_str(), B * HW, C, ms2,
           2.0 * B * (double)HW * C * 2 / ms2 / 1e6, st);
    HC(hipFree(x)); HC(hipFree(g)); HC(hipFree(b)); HC(hipFree(out)); HC(hipFree(sc));
}

// fused feed-forward (rowres.hip) against the three launches it replaces: LayerNorm, GEGLU projection, ff.net.2 + residual
static void bench_ff(const char* name, int M, int iters, Timer& t, void* zp) {
    if (!want(name)) return;
    const int C = 320;
    void* x = dalloc_bf16((size_t)M * C, 1);
    void* w1 = dalloc_bf16((size_t)8 * C * C, 2, 0.05f);
    void* w2 = dalloc_bf16((size_t)4 * C * C, 3, 0.03f);
    float* b1 = dalloc_f32(8 * C, 4);
    float* b2 = dalloc_f32(C, 5);
    float* lg = dalloc_f32(C, 6);
    float* lb = dalloc_f32(C, 7);
    void *st, *nb, *big, *out;
    HC(hipMalloc(&st, ff_stream_bytes(C)));
    HC(hipMalloc(&nb, (size_t)M * C * 2));
    HC(hipMalloc(&big, (size_t)M * 4 * C * 2));
    HC(hipMalloc(&out, (size_t)M * C * 2));
    int s0 = pack_ff_stream(w1, w2, st, C, 0);
    FFArgs a;
    a.x = x; a.out = out; a.ln_g = lg; a.ln_b = lb; a.stream = st; a.b1 = b1; a.b2 = b2; a.M = M; a.C = C;
    int s1 = DSIM_OK;
    const float msf = t.run([&] { s1 = launch_ff_fused(a, 0); }, iters);
    // interleaved rounds (rule 24): KB_FFSTAG=0,1,2,3,4 sweeps the wave de-phasing, KB_FFDBG=1,2,3 the ablation masks
    auto sweep = [&](const char* env, const char* label, int* knob, int restore) {
        const char* e = getenv(env);
        if (!e) return;
        std::vector<int> vals;
        std::string l = e;
        for (size_t pos = 0; pos < l.size();) {
            size_t nx = l.find(',', pos);
            if (nx == std::string::npos) nx = l.size();
            vals.push_back(atoi(l.substr(pos, nx - pos).c_str()));
            pos = nx + 1;
        }
        const int rounds = getenv("KB_ROUNDS") ? atoi(getenv("KB_ROUNDS")) : 5;
        std::vector<std::vector<float>> ms(vals.size());
        for (int r = 0; r < rounds; ++r)
            for (size_t k = 0; k < vals.size(); ++k) {
                *knob = vals[k];
                ms[k].push_back(t.run([&] { s1 |= launch_ff_fused(a, 0); }, iters));
            }
        *knob = restore;
        printf("  ff %s min/median ms:", label);
        for (size_t k = 0; k < vals.size(); ++k) {
            std::sort(ms[k].begin(), ms[k].end());
            printf("  %d:%.3f/%.3f", vals[k], ms[k][0], ms[k][rounds / 2]);
        }
        printf("\n");
    };
    sweep("KB_FFSTAG", "stagger", &g_ff_stagger, -1);
    sweep("KB_FFDBG", "ablation", &g_ff_dbg, 0);
    GemmArgs g1, g2;
    g1.A0 = nb; g1.C0 = C; g1.M = M; g1.N = 8 * C; g1.K = C; g1.W = w1; g1.bias = b1; g1.epi = EPI_GEGLU; g1.out = big; g1.ldo = 4 * C; g1.zero_page = zp;
    g2.A0 = big; g2.C0 = 4 * C; g2.M = M; g2.N = C; g2.K = 4 * C; g2.W = w2; g2.bias = b2; g2.epi = EPI_RESIDUAL; g2.residual = x; g2.out = out; g2.ldo = C; g2.zero_page = zp;
    g_force_bm = 0;
    const float msl = t.run([&] { s1 |= launch_layernorm(x, lg, lb, nb, M, C, 1e-5f, DSIM_BF16, 0); }, iters);
    const float ms1 = t.run([&] { s1 |= launch_gemm(g1, DSIM_BF16, 0); }, iters);
    const float ms2 = t.run([&] { s1 |= launch_gemm(g2, DSIM_BF16, 0); }, iters);
    const double fl = 2.0 * M * (double)C * 12 * C;
    printf("%-26s M=%7d  fused %7.3f ms %6.1f TF | ln %6.3f + geglu %6.3f + net2 %6.3f = %7.3f ms %6.1f TF  st=%d/%d\n", name, M, msf,
           fl / msf / 1e9, msl, ms1, ms2, msl + ms1 + ms2, fl / (msl + ms1 + ms2) / 1e9, s0, s1);
    HC(hipFree(x)); HC(hipFree(w1)); HC(hipFree(w2)); HC(hipFree(b1)); HC(hipFree(b2)); HC(hipFree(lg)); HC(hipFree(lb));
    HC(hipFree(st)); HC(hipFree(nb)); HC(hipFree(big)); HC(hipFree(out));
}

// row-resident Linear (rowres.hip) against the launches it replaces: [LayerNorm +] the 320-wide Linear (interleaved rounds)
static void bench_rowlin(const char* name, int M, int N, bool ln, int iters, Timer& t, void* zp) {
    if (!want(name)) return;
    const int C = 320;
    void* x = dalloc_bf16((size_t)M * C, 1);
    void* w = dalloc_bf16((size_t)N * C, 2, 0.05f);
    float* b = dalloc_f32(N, 4);
    float* lg = dalloc_f32(C, 6);
    float* lb = dalloc_f32(C, 7);
    void *st, *nb, *o1, *o2;
    float* dd;
    HC(hipMalloc(&st, rowlin_stream_bytes(C, N)));
    HC(hipMalloc(&nb, (size_t)M * C * 2));
    HC(hipMalloc(&o1, (size_t)M * N * 2));
    HC(hipMalloc(&o2, (size_t)M * N * 2));
    HC(hipMalloc(&dd, 4));
    HC(hipMemset(dd, 0, 4));
    int s0 = pack_rowlin_stream(w, st, C, N, 0), s1 = DSIM_OK;
    RowLinArgs a;
    a.x = x; a.out = o1; a.ln_g = ln ? lg : nullptr; a.ln_b = ln ? lb : nullptr; a.stream = st; a.M = M; a.C = C; a.N = N;
    GemmArgs g;
    g.A0 = ln ? nb : x; g.C0 = C; g.M = M; g.N = N; g.K = C; g.W = w; g.epi = EPI_NONE; g.out = o2; g.ldo = N; g.zero_page = zp;
    const int rounds = getenv("KB_ROUNDS") ? atoi(getenv("KB_ROUNDS")) : 5;
    std::vector<float> mf, ml, mg;
    for (int r = 0; r < rounds; ++r) {
        mf.push_back(t.run([&] { s1 |= launch_rowlin(a, 0); }, iters));
        if (ln) ml.push_back(t.run([&] { s1 |= launch_layernorm(x, lg, lb, nb, M, C, 1e-5f, DSIM_BF16, 0); }, iters));
        mg.push_back(t.run([&] { s1 |= launch_gemm(g, DSIM_BF16, 0); }, iters));
    }
    if (const char* e = getenv("KB_RLDBG")) {       // ablation masks, interleaved with the full kernel
        std::vector<int> vals{0};
        std::string l = e;
        for (size_t pos = 0; pos < l.size();) {
            size_t nx = l.find(',', pos);
            if (nx == std::string::npos) nx = l.size();
            vals.push_back(atoi(l.substr(pos, nx - pos).c_str()));
            pos = nx + 1;
        }
        std::vector<std::vector<float>> ms(vals.size());
        for (int r = 0; r < rounds; ++r)
            for (size_t k = 0; k < vals.size(); ++k) {
                g_rl_dbg = vals[k];
                ms[k].push_back(t.run([&] { s1 |= launch_rowlin(a, 0); }, iters));
            }
        g_rl_dbg = 0;
        printf("  rowlin ablation median ms:");
        for (size_t k = 0; k < vals.size(); ++k) {
            std::sort(ms[k].begin(), ms[k].end());
            printf("  %d:%.3f", vals[k], ms[k][rounds / 2]);
        }
        printf("\n");
        s1 |= launch_rowlin(a, 0);
    }
    hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, (const __bf16*)o1, (const __bf16*)o2, (size_t)M * N, dd);
    float d = 0.f;
    HC(hipMemcpy(&d, dd, 4, hipMemcpyDeviceToHost));
    std::sort(mf.begin(), mf.end()); std::sort(mg.begin(), mg.end());
    float lnm = 0.f;
    if (ln) { std::sort(ml.begin(), ml.end()); lnm = ml[rounds / 2]; }
    const double fl = 2.0 * M * (double)C * N;
    for (int wpc = 1; wpc <= 3; ++wpc) {
        g_rl_wpc = wpc;
        std::vector<float> qa;
        for (int r = 0; r < 5; ++r) qa.push_back(t.run([&] { s1 |= launch_rowlin(a, 0); }, iters));
        std::sort(qa.begin(), qa.end());
        printf("  rowlin with %d workgroups per CU: %7.3f ms\n", wpc, qa[2]);
    }
    g_rl_wpc = 3;
    printf("%-26s M=%7d  rowlin %7.3f ms %6.1f TF | ln %6.3f + gemm %6.3f = %7.3f ms   maxdiff %.4g  st=%d/%d\n", name, M, mf[rounds / 2],
           fl / mf[rounds / 2] / 1e9, lnm, mg[rounds / 2], lnm + mg[rounds / 2], d, s0, s1);
    HC(hipFree(x)); HC(hipFree(w)); HC(hipFree(b)); HC(hipFree(lg)); HC(hipFree(lb));
    HC(hipFree(st)); HC(hipFree(nb)); HC(hipFree(o1)); HC(hipFree(o2)); HC(hipFree(dd));
}

int main(int argc, char** argv) {
    const int B2 = argc > 1 ? atoi(argv[1]) : 64;
    const int iters = argc > 2 ? atoi(argv[2]) : 10;
    g_filter = argc > 3 ? argv[3] : nullptr;
    if (const char* e = getenv("KB_ONEEXP")) g_gemm_exp = atoi(e);      // one GEMM experiment mask for the whole run (PMC passes)
    void* zp;
    HC(hipMalloc(&zp, 256));
    HC(hipMemset(zp, 0, 256));
    Timer t;
    const int s64 = B2 * 4096, s32 = B2 * 1024, s16 = B2 * 256, s8 = B2 * 64;
    bench_ff("ff_64_320_fused", s64, iters, t, zp);
    bench_rowlin("rowlin_64_ln_qkv", s64, 960, true, iters, t, zp);
    bench_rowlin("rowlin_64_ln_q", s64, 320, true, iters, t, zp);
    bench_rowlin("rowlin_64_q", s64, 320, false, iters, t, zp);
    // ---- 3x3 convs ----
    bench_gemm("conv3_64_320_320", GEMM_CONV3, s64, 320, 320, 64, 64, EPI_NONE, iters, t, zp);
    bench_gemm("conv3_64_320_320_res", GEMM_CONV3, s64, 320, 320, 64, 64, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("conv3_32_320_640", GEMM_CONV3, s32, 640, 320, 32, 32, EPI_NONE, iters, t, zp);
    bench_gemm("conv3_32_640_640", GEMM_CONV3, s32, 640, 640, 32, 32, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("conv3_16_640_1280", GEMM_CONV3, s16, 1280, 640, 16, 16, EPI_NONE, iters, t, zp);
    bench_gemm("conv3_16_1280_1280", GEMM_CONV3, s16, 1280, 1280, 16, 16, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("conv3_16_2560_1280", GEMM_CONV3, s16, 1280, 2560, 16, 16, EPI_NONE, iters, t, zp);
    bench_gemm("conv3_8_1280_1280", GEMM_CONV3, s8, 1280, 1280, 8, 8, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("conv3_8_2560_1280", GEMM_CONV3, s8, 1280, 2560, 8, 8, EPI_NONE, iters, t, zp);
    // ---- the VAE encoder's widest levels (pixels-in path): B2 / 16 images at 512 px ----
    {
        const int vi = std::max(1, B2 / 16);
        bench_gemm("conv3_vae512_128_128", GEMM_CONV3, vi * 512 * 512, 128, 128, 512, 512, EPI_NONE, iters, t, zp);
        bench_gemm("conv3_vae512_128_128_res", GEMM_CONV3, vi * 512 * 512, 128, 128, 512, 512, EPI_RESIDUAL, iters, t, zp);
        bench_gemm("conv3_vae256_128_256", GEMM_CONV3, vi * 256 * 256, 256, 128, 256, 256, EPI_NONE, iters, t, zp);
        bench_gemm("conv3_vae256_256_256_res", GEMM_CONV3, vi * 256 * 256, 256, 256, 256, 256, EPI_RESIDUAL, iters, t, zp);
        bench_gemm("conv3_vae128_512_512_res", GEMM_CONV3, vi * 128 * 128, 512, 512, 128, 128, EPI_RESIDUAL, iters, t, zp);
    }
    // ---- linears ----
    bench_gemm("lin_64_320_320_res", GEMM_LINEAR, s64, 320, 320, 0, 0, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("lin_64_320_960_qkv", GEMM_LINEAR, s64, 960, 320, 0, 0, EPI_NONE, iters, t, zp);
    bench_gemm("lin_64_320_2560_geglu", GEMM_LINEAR, s64, 2560, 320, 0, 0, EPI_GEGLU, iters, t, zp);
    bench_gemm("lin_64_1280_320_res", GEMM_LINEAR, s64, 320, 1280, 0, 0, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("lin_32_640_640_res", GEMM_LINEAR, s32, 640, 640, 0, 0, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("lin_32_640_1920_qkv", GEMM_LINEAR, s32, 1920, 640, 0, 0, EPI_NONE, iters, t, zp);
    bench_gemm("lin_32_640_5120_geglu", GEMM_LINEAR, s32, 5120, 640, 0, 0, EPI_GEGLU, iters, t, zp);
    bench_gemm("lin_32_2560_640_res", GEMM_LINEAR, s32, 640, 2560, 0, 0, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("lin_16_1280_1280_res", GEMM_LINEAR, s16, 1280, 1280, 0, 0, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("lin_16_1280_3840_qkv", GEMM_LINEAR, s16, 3840, 1280, 0, 0, EPI_NONE, iters, t, zp);
    bench_gemm("lin_16_1280_10240_geglu", GEMM_LINEAR, s16, 10240, 1280, 0, 0, EPI_GEGLU, iters, t, zp);
    bench_gemm("lin_16_1280_10240_plain", GEMM_LINEAR, s16, 10240, 1280, 0, 0, EPI_NONE, iters, t, zp);
    bench_gemm("lin_32_640_5120_plain", GEMM_LINEAR, s32, 5120, 640, 0, 0, EPI_NONE, iters, t, zp);
    bench_gemm("lin_16_5120_1280_res", GEMM_LINEAR, s16, 1280, 5120, 0, 0, EPI_RESIDUAL, iters, t, zp);
    bench_gemm("lin_16_sc_2560_1280", GEMM_LINEAR, s16, 1280, 1280, 0, 0, EPI_NONE, iters, t, zp, 1280);
    bench_gemm("lin_kv_768_2560", GEMM_LINEAR, 154, 2560, 768, 0, 0, EPI_NONE, iters, t, zp);
    // ---- attention ----
    // DiT-XL/2 at 256 px: 256 tokens per image, hidden 1152 (B2 = 256 elements <-> 64 pairs, both CFG halves)
    bench_gemm("lin_dit_qkv", GEMM_LINEAR, B2 * 256, 3456, 1152, 0, 0, EPI_NONE, iters, t, zp);
    bench_gemm("lin_dit_fc1_act", GEMM_LINEAR, B2 * 256, 4608, 1152, 0, 0, EPI_NONE, iters, t, zp, 0, 1);
    bench_gemm("lin_dit_proj_gate_res", GEMM_LINEAR, B2 * 256, 1152, 1152, 0, 0, EPI_RESIDUAL, iters, t, zp, 0, 0, true);
    bench_gemm("lin_dit_fc2_gate_res", GEMM_LINEAR, B2 * 256, 1152, 4608, 0, 0, EPI_RESIDUAL, iters, t, zp, 0, 0, true);
    bench_gemm("lin_dit_proj_res_nogate", GEMM_LINEAR, B2 * 256, 1152, 1152, 0, 0, EPI_RESIDUAL, iters, t, zp);      // the same shapes without the adaLN gate
    bench_gemm("lin_dit_fc2_res_nogate", GEMM_LINEAR, B2 * 256, 1152, 4608, 0, 0, EPI_RESIDUAL, iters, t, zp);
    bench_attn("attn_self_4096_d40", B2, B2, 8, 4096, 4096, 40, iters, t);
    bench_attn("attn_self_1024_d80", B2, B2, 8, 1024, 1024, 80, iters, t);
    bench_attn("attn_self_256_d160", B2, B2, 8, 256, 256, 160, iters, t);
    bench_attn("attn_cross_4096_d40", B2, 2, 8, 4096, 77, 40, iters, t);
    bench_attn("attn_cross_1024_d80", B2, 2, 8, 1024, 77, 80, iters, t);
    bench_attn("attn_cross_256_d160", B2, 2, 8, 256, 77, 160, iters, t);
    bench_tail("tail_256_d160", std::max(1, B2 / 4), iters, t);
    bench_tail("tail_256_d160_1pair", 1, iters, t);
    bench_attn("attn_dit_256_d72", B2, B2, 16, 256, 256, 72, iters, t);
    bench_attn("attn_sdxl_self_4096_d64", B2, B2, 10, 4096, 4096, 64, iters, t);
    bench_attn("attn_sdxl_self_1024_d64", B2, B2, 20, 1024, 1024, 64, iters, t);
    bench_attn("attn_sdxl_cross_4096_d64", B2, 2, 10, 4096, 77, 64, iters, t);
    bench_attn("attn_sdxl_cross_1024_d64", B2, 2, 20, 1024, 77, 64, iters, t);
    // ---- norms ----
    bench_gn("gn_64_320", B2, 4096, 320, iters, t);
    bench_gn("gn_32_640", B2, 1024, 640, iters, t);
    bench_gn("gn_16_1280", B2, 256, 1280, iters, t);
    bench_gn("gn_vae512_128", std::max(1, B2 / 16), 512 * 512, 128, iters, t);       // the VAE's levels (B2 / 16 images)
    bench_gn("gn_vae256_256", std::max(1, B2 / 16), 256 * 256, 256, iters, t);
    bench_gn("gn_vae128_512", std::max(1, B2 / 16), 128 * 128, 512, iters, t);
    return 0;
}
