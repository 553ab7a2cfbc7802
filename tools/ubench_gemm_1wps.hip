// Development micro-benchmark (gfx950): the K loop of the 256 x 320 implicit-GEMM tile in three structures, with REAL LDS-DMA
// staging from global memory, real swizzled ds_read_b128 fragment reads, one barrier per K tile and no epilogue -- the measurement
// the round-4 review asked for before a one-wave-per-SIMD conv / GEMM kernel is built.
//   VAR 0  the shipped structure: 8 waves as 4 x 2, 64 x 160 per wave (160 accumulator registers, 2 waves per SIMD), fragment
//          reads pinned one piece ahead, nine DMA pieces per wave per K tile in a burst behind the barrier (gemm.hip's loop)
//   VAR 1  4 waves as 2 x 2, 128 x 160 per wave, ONE wave per SIMD: 320 accumulator registers (256 in AGPRs + 64 in VGPRs, the
//          MFMAs in inline asm because hipcc picks ONE accumulator class per function), fragment reads 0.225 instead of 0.35 KB
//          per MFMA, a 4-deep weight-fragment ring read two groups ahead, 18 DMA pieces per wave dealt over the MFMA groups
//   VAR 2  VAR 1 with the WEIGHT operand direct to registers from a fragment-order pack (1 KB contiguous per wave instruction,
//          buffer_load_dwordx4, one K step ahead) -- only the activations go through LDS (a 4-stage ring, counted vmcnt)
// Problem: A [65536 rows][K] bf16 (tile t of workgroup b reads rows ((16 b + t) % 256) * 256 ...: every row block is re-read by
// 16 tiles, as a 3x3 conv's taps re-read the activations), W [320][K]; each workgroup walks `tiles` tiles of K / 64 K tiles.
// Prints ms, TF/s, in-kernel clock and a checksum of the accumulators per variant (the three must agree).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gemm_1wps.hip -o tools/ubench_gemm_1wps ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 320, STAGE = (BM + BN) * 128;
constexpr unsigned OOB = 0x80000000u;

struct P {
    const void* A; const void* W; const void* Wp;    // Wp: fragment-order pack of W (VAR 2)
    float* out; long long* clk;
    int K, tiles, arows;
    unsigned a_bytes, w_bytes;
};

// ---- VAR 0: gemm.hip's K loop (WM = 4, WN = 2, TM = 4, TN = 10, PS = 5, NP = 2) -----------------------------------------------
__global__ __launch_bounds__(512, 2) void k_var0(const P p) {
    constexpr int NW = 8, WN = 2, TM = 4, TN = 10, PS = 5, NP = 2, NA = 4, NB = 5, A_BYTES = BM * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 3, wrow = wave * 8 + lrow;
    const unsigned celb = ((lane & 7) ^ ((wrow >> 1) & 7)) * 16;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
    unsigned a_voff[NA], b_voff[NB];
    auto setup = [&](int tile) {
        const int m0 = ((blockIdx.x * 16 + tile) % (p.arows / BM)) * BM;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_voff[i] = (unsigned)(m0 + i * 64 + wrow) * (unsigned)p.K * 2u + celb;
#pragma unroll
        for (int i = 0; i < NB; ++i) b_voff[i] = (unsigned)(i * 64 + wrow) * (unsigned)p.K * 2u + celb;
    };
    auto issue = [&](int t, int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
        const int soff = t * 128;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sa + (i * NW + wave_u) * 1024), 16, (int)a_voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(sb + (i * NW + wave_u) * 1024), 16, (int)b_voff[i], soff, 0, 0);
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / 64;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int quad = lane >> 4;
    const int foff0 = (lane & 15) * 128 + ((quad ^ ((lane >> 1) & 7)) << 4);
    bf16x8 xf[2][TM], wf[2][PS];
    auto load_x = [&](int buf, int step, int set) {
        const char* sa = smem + buf * STAGE + wm * 64 * 128 + (foff0 ^ (step << 6));
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[set][i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 128);
    };
    auto load_w = [&](int buf, int q, int set) {
        const int step = q / NP, j0 = (q - step * NP) * PS;
        const char* sb = smem + buf * STAGE + A_BYTES + (wn * 160 + j0 * 16) * 128 + (foff0 ^ (step << 6));
#pragma unroll
        for (int j = 0; j < PS; ++j) wf[set][j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 128);
    };
    auto mma_piece = [&](int q) {
        const int step = q / NP, j0 = (q - step * NP) * PS;
#pragma unroll
        for (int j = 0; j < PS; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q & 1][j], xf[step][i], acc[i][j0 + j], 0, 0, 0);
    };
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int b0 = 0;
    setup(0);
    issue(0, 0);
    for (int tile = 0; tile < p.tiles; ++tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        load_x(b0, 0, 0);
        load_w(b0, 0, 0);
        for (int t = 0; t < nk; ++t) {
            const int cur = b0 ^ (t & 1);
            if (t + 1 < nk) issue(t + 1, cur ^ 1);
#pragma unroll
            for (int q = 0; q + 1 < 2 * NP; ++q) {
                if (q == 0) load_x(cur, 1, 1);
                load_w(cur, q + 1, (q + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_piece(q);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 1 < nk) { load_x(cur ^ 1, 0, 0); load_w(cur ^ 1, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
            mma_piece(2 * NP - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int xbuf = b0 ^ ((nk - 1) & 1);
        if (tile + 1 < p.tiles) { setup(tile + 1); issue(0, xbuf ^ 1); }
        b0 = xbuf ^ 1;
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    atomicAdd(p.out + blockIdx.x, s);
    if (lane == 0) { p.clk[(blockIdx.x * 8 + wave) * 2] = t1 - t0; p.clk[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

// ---- VAR 3: TWO independent 4-wave workgroups per CU (round 6): 128 x 320 tiles as 2 x 2 waves of 64 x 160 -- the shipped wave tile --
// with 64-byte LDS rows (one MFMA K step per stage, 2 stages = 56 KB per workgroup).  The two waves of a SIMD then belong to
// different workgroups: they do not meet at one barrier, and (in a real kernel) one workgroup's epilogue runs under the other's K loop.
// 16-row DMA pieces; the 16-byte chunk of row r sits at chunk c ^ f(r >> 2), f = (0, 2, 3, 1): conflict-free ds_read_b128 groups.
__global__ __launch_bounds__(256, 2) void k_var3(const P p) {
    constexpr int NW = 4, BM3 = 128, TM = 4, TN = 10, PS = 5, A_BYTES = BM3 * 64, STG = (BM3 + BN) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane >> 2;
    const unsigned lc = (unsigned)(lane & 3) ^ ((0x78u >> (2 * (lr >> 2))) & 3u);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
    unsigned a_voff[2], b_voff[5];
    auto setup = [&](int tile) {
        const int m0 = (int)(((long)blockIdx.x * 16 + tile) % (p.arows / BM3)) * BM3;
#pragma unroll
        for (int i = 0; i < 2; ++i) a_voff[i] = (unsigned)(m0 + (i * NW + wave) * 16 + lr) * (unsigned)p.K * 2u + lc * 16u;
#pragma unroll
        for (int i = 0; i < 5; ++i) b_voff[i] = (unsigned)((i * NW + wave) * 16 + lr) * (unsigned)p.K * 2u + lc * 16u;
    };
    auto issue = [&](int t, int buf) {
        char* sa = smem + buf * STG;
        char* sb = sa + A_BYTES;
        const int soff = t * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sa + (i * NW + wave) * 1024), 16, (int)a_voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(sb + (i * NW + wave) * 1024), 16, (int)b_voff[i], soff, 0, 0);
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / 32;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = fr * 64 + ((fq ^ ((0x78 >> (2 * (fr >> 2))) & 3)) << 4);
    bf16x8 xf[2][TM], wf[2][PS];
    auto load_x = [&](int buf, int set) {
        const char* sa = smem + buf * STG + wm * 64 * 64 + foff;
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[set][i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 64);
    };
    auto load_w = [&](int buf, int piece, int set) {
        const char* sb = smem + buf * STG + A_BYTES + (wn * 160 + piece * PS * 16) * 64 + foff;
#pragma unroll
        for (int j = 0; j < PS; ++j) wf[set][j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 64);
    };
    auto mma_piece = [&](int piece, int xs) {
#pragma unroll
        for (int j = 0; j < PS; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i][piece * PS + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[piece][j], xf[xs][i], acc[i][piece * PS + j], 0, 0, 0);
    };
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int b0 = 0;
    setup(0);
    issue(0, 0);
    for (int tile = 0; tile < p.tiles; ++tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        load_x(b0, 0);
        load_w(b0, 0, 0);
        for (int t = 0; t < nk; t += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {          // (two K steps per trip: the fragment sets alternate at compile time)
                const int cur = b0 ^ u;
                if (t + u + 1 < nk) issue(t + u + 1, cur ^ 1);
                load_w(cur, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_piece(0, u);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t + u + 1 < nk) { load_x(cur ^ 1, u ^ 1); load_w(cur ^ 1, 0, 0); }
                __builtin_amdgcn_sched_barrier(0);
                mma_piece(1, u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // nk is even: the last step read buffer b0 ^ 1; the next tile's first stage goes to b0 and is read with set 0
        if (tile + 1 < p.tiles) { setup(tile + 1); issue(0, b0); }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    atomicAdd(p.out + (blockIdx.x & 255), s);
    if (lane == 0) { p.clk[(blockIdx.x * 4 + wave) * 2] = t1 - t0; p.clk[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

// ---- VAR 1 / VAR 2: one wave per SIMD, 128 x 160 per wave ---------------------------------------------------------------------
// eight MFMAs that share one weight fragment: acc[i] += w x x[i]; accumulators in AGPRs ("a") or VGPRs ("v").  The leading s_nop
// covers a VGPR written by the instruction in front of the statement (hipcc pads nothing for an asm statement).
#define MFMA8(C, w, x, a0, a1, a2, a3, a4, a5, a6, a7)                                                                         \
    asm volatile("s_nop 1\n\t"                                                                                                 \
                 "v_mfma_f32_16x16x32_bf16 %0, %8, %9, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %8, %10, %1\n\t"                      \
                 "v_mfma_f32_16x16x32_bf16 %2, %8, %11, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %8, %12, %3\n\t"                     \
                 "v_mfma_f32_16x16x32_bf16 %4, %8, %13, %4\n\tv_mfma_f32_16x16x32_bf16 %5, %8, %14, %5\n\t"                     \
                 "v_mfma_f32_16x16x32_bf16 %6, %8, %15, %6\n\tv_mfma_f32_16x16x32_bf16 %7, %8, %16, %7"                         \
                 : "+" C(a0), "+" C(a1), "+" C(a2), "+" C(a3), "+" C(a4), "+" C(a5), "+" C(a6), "+" C(a7)                      \
                 : "v"(w), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]))

template <int VAR, int PPG>
__global__ __launch_bounds__(256, 1) void k_var12(const P p) {
    constexpr int NW = 4, A_BYTES = BM * 128;
    constexpr int ASTG = VAR == 2 ? 4 : 2;                     // LDS ring depth
    constexpr int SB = VAR == 2 ? A_BYTES : STAGE;             // bytes per ring stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, wrow = wave * 8 + lrow;
    const unsigned celb = ((lane & 7) ^ ((wrow >> 1) & 7)) * 16;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rWp = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (int)p.w_bytes, 0x00020000);
    const int nk = p.K / 64;
    const int wm = wave >> 1, wn = wave & 1;
    const int quad = lane >> 4;
    const int foff0 = (lane & 15) * 128 + ((quad ^ ((lane >> 1) & 7)) << 4);

    // DMA piece d (0..17; VAR 2: 0..7, activations only) of K tile t of the tile whose first row is m0, into ring stage `buf`
    auto piece = [&](int m0, int d, int t, int buf) {
        char* s = smem + buf * SB;
        if (d < 8) {
            const unsigned voff = (unsigned)(m0 + d * 32 + wrow) * (unsigned)p.K * 2u + celb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(s + (d * NW + wave) * 1024), 16, (int)voff, t * 128, 0, 0);
        } else {
            const int i = d - 8;
            const unsigned voff = (unsigned)(i * 32 + wrow) * (unsigned)p.K * 2u + celb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(s + A_BYTES + (i * NW + wave) * 1024), 16, (int)voff, t * 128, 0, 0);
        }
    };

    f32x4 aa[8][8];          // accumulator tiles (i, j < 8): AGPRs
    f32x4 av[8][2];          // accumulator tiles (i, j = 8, 9): VGPRs
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) aa[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        av[i][0] = av[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 xf[2][8];         // activation fragments of the two K steps of a tile
    bf16x8 wf[4];            // VAR 1: weight-fragment ring, group g in slot g & 3, read two groups ahead
    auto rd_x = [&](int buf, int step, int i) {
        xf[step][i] = *reinterpret_cast<const bf16x8*>(smem + buf * SB + (wm * 128 + i * 16) * 128 + (foff0 ^ (step << 6)));
    };
    auto rd_w = [&](int buf, int g) {      // group g = 10 step + j
        const int step = g / 10, j = g - step * 10;
        wf[g & 3] = *reinterpret_cast<const bf16x8*>(smem + buf * SB + A_BYTES + (wn * 160 + j * 16) * 128 + (foff0 ^ (step << 6)));
    };
    auto group = [&](int g, const bf16x8& w) {
        const int step = g / 10, j = g - step * 10;
        if (j < 8) {
            MFMA8("a", w, xf[step], aa[0][j], aa[1][j], aa[2][j], aa[3][j], aa[4][j], aa[5][j], aa[6][j], aa[7][j]);
        } else {
            MFMA8("v", w, xf[step], av[0][j - 8], av[1][j - 8], av[2][j - 8], av[3][j - 8], av[4][j - 8], av[5][j - 8], av[6][j - 8], av[7][j - 8]);
        }
    };
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    const int nmt = p.arows / BM;
    if constexpr (VAR == 1) {
        int m0 = ((blockIdx.x * 16) % nmt) * BM;
#pragma unroll
        for (int d = 0; d < 18; ++d) piece(m0, d, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) rd_x(0, 0, i);
        rd_w(0, 0); rd_w(0, 1);
        const int total = p.tiles * nk;
        int tile = 0, t = 0;
        for (int it = 0; it < total; ++it) {
            const int cur = it & 1;
            // the K tile staged during this one: (tile, t + 1) or (tile + 1, 0) (the walk's last K tile stages one nobody reads)
            int nt = t + 1, nm0 = m0;
            if (nt == nk) { nt = 0; ++tile; nm0 = ((blockIdx.x * 16 + tile) % nmt) * BM; }
#pragma unroll
            for (int g = 0; g < 18; ++g) {
                rd_w(cur, g + 2);
                if (g < 8) rd_x(cur, 1, g);                     // the second K step's activation fragments, one per group
#pragma unroll
                for (int d = 0; d < 18; ++d) if (d / PPG == g) piece(nm0, d, nt, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                group(g, wf[g & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // every fragment of this stage is in registers; the next stage has landed in every wave behind the barrier
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) rd_x(cur ^ 1, 0, i);
            rd_w(cur ^ 1, 0); rd_w(cur ^ 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            group(18, wf[2]);
            group(19, wf[3]);
            __builtin_amdgcn_sched_barrier(0);
            t = nt; m0 = nm0;
        }
    } else {
        // VAR 2: weights from the fragment-order pack, one K step (ten 1-KB loads) ahead; activations through a 4-stage ring,
        // K tile it + 2 staged during K tile it.  vmcnt is counted: per K step a wave issues [10 weight loads, 4 DMA pieces];
        // waiting for the previous step's weight loads = all but the 4 pieces behind them (and every older piece has landed).
        bf16x8 wq[2][10];
        auto ld_w = [&](int t, int step, int set) {         // K step 2 t + step of a tile (the pack is the same for every tile)
            const int base = ((wn * (2 * nk) + 2 * t + step) * 10) * 1024 + lane * 16;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                u32x4 v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(base + j * 1024), "s"(rWp) : "memory");
                wq[set][j] = __builtin_bit_cast(bf16x8, v);
            }
        };
        const int total = p.tiles * nk;
        // (tile, t) of K tile it, it + 1, it + 2 of the walk, kept incrementally
        int t0k = 0, t1k = nk > 1 ? 1 : 0, tl1 = nk > 1 ? 0 : 1;
        int t2k = t1k + 1, tl2 = tl1;
        if (t2k == nk) { t2k = 0; ++tl2; }
        const int mb = blockIdx.x * 16;
        // prologue: stages of K tiles 0 and 1, weights of step 0
#pragma unroll
        for (int d = 0; d < 8; ++d) piece((mb % nmt) * BM, d, 0, 0);
#pragma unroll
        for (int d = 0; d < 8; ++d) piece(((mb + tl1) % nmt) * BM, d, t1k, 1);
        ld_w(0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) rd_x(0, 0, i);
        for (int it = 0; it < total; ++it) {
            const int cur = it & 3;
            const int m2 = ((mb + tl2) % nmt) * BM;
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                // weights of the NEXT K step, then this step's share of the DMA pieces of K tile it + 2
                if (step == 0) ld_w(t0k, 1, 1); else ld_w(t1k, 0, 0);
#pragma unroll
                for (int d = 0; d < 4; ++d) piece(m2, step * 4 + d, t2k, (it + 2) & 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    if (step == 0 && j < 8) rd_x(cur, 1, j);
                    if (step == 1 && j == 7) {
                        // the next K tile's stage: its pieces were issued one K tile ago; every wave is past them at this barrier
                        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    if (step == 1 && j >= 8) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rd_x((it + 1) & 3, 0, (j - 8) * 4 + i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    group(step * 10 + j, wq[step][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the weight loads issued at the top of this step are the next step's operands: all but the 4 pieces behind them
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 10; ++j) asm volatile("" : "+v"(wq[step ^ 1][j]));
            }
            t0k = t1k; t1k = t2k;
            if (++t2k == nk) { t2k = 0; ++tl2; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += aa[i][j][0] + aa[i][j][1] + aa[i][j][2] + aa[i][j][3];
        s += av[i][0][0] + av[i][0][1] + av[i][0][2] + av[i][0][3] + av[i][1][0] + av[i][1][1] + av[i][1][2] + av[i][1][3];
    }
    atomicAdd(p.out + blockIdx.x, s);
    if (lane == 0) { p.clk[(blockIdx.x * 8 + wave) * 2] = t1 - t0; p.clk[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

__global__ void fill_bf16(__bf16* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (__bf16)(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale);
}
// fragment-order pack of W for VAR 2: [wn][k step][j][lane][8 elements]: lane (r = lane & 15, q = lane >> 4) of fragment j of
// column half wn holds W[wn * 160 + 16 j + r][32 ks + 8 q ..]
__global__ void pack_w(const __bf16* w, __bf16* wp, int K) {
    const int nks = K / 32;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte unit
    if (i >= (size_t)2 * nks * 10 * 64) return;
    const int lane = i & 63;
    size_t r = i >> 6;
    const int j = r % 10; r /= 10;
    const int ks = r % nks; const int wn = (int)(r / nks);
    const int row = wn * 160 + 16 * j + (lane & 15), k = 32 * ks + 8 * (lane >> 4);
    *reinterpret_cast<u32x4*>(wp + i * 8) = *reinterpret_cast<const u32x4*>(w + (size_t)row * K + k);
}

#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2880;
    const int tiles = argc > 2 ? atoi(argv[2]) : 16;
    const int rounds = argc > 3 ? atoi(argv[3]) : 7;
    const int nb = 256, arows = 65536;
    P p{};
    void *A, *W, *Wp;
    HC(hipMalloc(&A, (size_t)arows * K * 2)); HC(hipMalloc(&W, (size_t)BN * K * 2)); HC(hipMalloc(&Wp, (size_t)BN * K * 2));
    hipLaunchKernelGGL(fill_bf16, dim3((unsigned)(((size_t)arows * K + 255) / 256)), dim3(256), 0, 0, (__bf16*)A, (size_t)arows * K, 1u, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3((unsigned)(((size_t)BN * K + 255) / 256)), dim3(256), 0, 0, (__bf16*)W, (size_t)BN * K, 2u, 0.05f);
    hipLaunchKernelGGL(pack_w, dim3((unsigned)((2 * (K / 32) * 10 * 64 + 255) / 256)), dim3(256), 0, 0, (const __bf16*)W, (__bf16*)Wp, K);
    HC(hipMalloc((void**)&p.out, nb * 4)); HC(hipMalloc((void**)&p.clk, nb * 8 * 2 * 8));
    p.A = A; p.W = W; p.Wp = Wp; p.K = K; p.tiles = tiles; p.arows = arows;
    p.a_bytes = (unsigned)((size_t)arows * K * 2); p.w_bytes = (unsigned)((size_t)BN * K * 2);
    const int lds0 = 2 * STAGE, lds2 = 4 * BM * 128;
    auto k11 = k_var12<1, 1>; auto k12 = k_var12<1, 2>; auto k13 = k_var12<1, 3>; auto k22 = k_var12<2, 2>;
    HC(hipFuncSetAttribute((const void*)k_var0, hipFuncAttributeMaxDynamicSharedMemorySize, lds0));
    HC(hipFuncSetAttribute((const void*)k11, hipFuncAttributeMaxDynamicSharedMemorySize, lds0));
    HC(hipFuncSetAttribute((const void*)k12, hipFuncAttributeMaxDynamicSharedMemorySize, lds0));
    HC(hipFuncSetAttribute((const void*)k13, hipFuncAttributeMaxDynamicSharedMemorySize, lds0));
    HC(hipFuncSetAttribute((const void*)k22, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
    hipEvent_t a, b; HC(hipEventCreate(&a)); HC(hipEventCreate(&b));
    const double flop = 2.0 * BM * BN * (double)K * tiles * nb;
    // variants: 0 | 1 with 1, 2, 3 pieces per group | 2
    struct V { int var, ppg; const char* name; };
    const V vs[] = {{0, 0, "VAR0 8 waves 64x160 (shipped loop)"}, {1, 1, "VAR1 4 waves 128x160, 1 DMA piece / group"},
                    {1, 2, "VAR1 4 waves 128x160, 2 DMA pieces / group"}, {1, 3, "VAR1 4 waves 128x160, 3 DMA pieces / group"},
                    {2, 0, "VAR2 4 waves 128x160, weights direct to registers"},
                    {3, 0, "VAR3 2 workgroups x 4 waves per CU, 128x320, 64-B rows"}};
    const int nv = sizeof(vs) / sizeof(vs[0]);
    std::vector<std::vector<float>> ms(nv);
    std::vector<std::vector<double>> ghz(nv);
    std::vector<double> chk(nv, 0.0);
    for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < nv; ++v) {
            HC(hipMemset(p.out, 0, nb * 4));
            HC(hipEventRecord(a, 0));
            if (vs[v].var == 3) hipLaunchKernelGGL(k_var3, dim3(2 * nb), dim3(256), 2 * (128 + BN) * 64, 0, p);
            else if (vs[v].var == 0) hipLaunchKernelGGL(k_var0, dim3(nb), dim3(512), lds0, 0, p);
            else if (vs[v].var == 1 && vs[v].ppg == 1) hipLaunchKernelGGL(k11, dim3(nb), dim3(256), lds0, 0, p);
            else if (vs[v].var == 1 && vs[v].ppg == 2) hipLaunchKernelGGL(k12, dim3(nb), dim3(256), lds0, 0, p);
            else if (vs[v].var == 1) hipLaunchKernelGGL(k13, dim3(nb), dim3(256), lds0, 0, p);
            else hipLaunchKernelGGL(k22, dim3(nb), dim3(256), lds2, 0, p);
            HC(hipEventRecord(b, 0)); HC(hipEventSynchronize(b));
            HC(hipGetLastError());
            float t; HC(hipEventElapsedTime(&t, a, b));
            const int nwv = vs[v].var == 0 ? 8 : 4;
            std::vector<long long> hc(nb * 8 * 2);
            HC(hipMemcpy(hc.data(), p.clk, hc.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> g;
            if (vs[v].var == 3) { for (int i = 0; i < 2 * nb * 4; ++i) g.push_back((double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1); }
            else for (int i = 0; i < nb; ++i) for (int w = 0; w < nwv; ++w) g.push_back((double)hc[2 * (i * 8 + w)] / (double)hc[2 * (i * 8 + w) + 1] * 0.1);
            std::sort(g.begin(), g.end());
            std::vector<float> ho(nb);
            HC(hipMemcpy(ho.data(), p.out, nb * 4, hipMemcpyDeviceToHost));
            double c = 0; for (float x : ho) c += x;
            chk[v] = c;
            if (r) { ms[v].push_back(t); ghz[v].push_back(g[g.size() / 2]); }
        }
    printf("K = %d, %d tiles per workgroup, %d workgroups; %.1f GFLOP per launch\n", K, tiles, nb, flop / 1e9);
    for (int v = 0; v < nv; ++v) {
        std::sort(ms[v].begin(), ms[v].end()); std::sort(ghz[v].begin(), ghz[v].end());
        const int m = (int)ms[v].size() / 2;
        const double mf = ms[v][m] * 1e-3 * ghz[v][m] * 1e9 / ((double)(K / 32) * tiles * 160.0);    // cycles per MFMA slot of a SIMD (16 = pipe-bound)
        printf("%-52s min %.3f median %.3f ms  %7.1f TF/s  clock %.3f GHz  cycles per MFMA %.2f  checksum %.6e\n", vs[v].name, ms[v][0], ms[v][m],
               flop / ms[v][m] / 1e9, ghz[v][m], mf, chk[v]);
    }
    return 0;
}
