// Row-resident transformer kernels for the 320-channel (64 x 64) level of the SD U-Net, gfx950, h16.
//
// The BasicTransformerBlock's feed-forward at C = 320 -- LayerNorm -> Linear(C, 8C) -> h * gelu(g) -> Linear(4C, C) ->
// + residual (control flow: /root/reference/diffsim/hacked_modules.py:82-132, the diffusers FeedForward / GEGLU modules
// behind it) -- is ONE kernel here.  As three launches it moved the 4C-wide intermediate through HBM twice (1.34 GB
// written + 1.34 GB read per layer at 32 pairs) and spent as long in the erf-GELU epilogue as in the MFMAs.
//
// Design: a wave owns 32 token rows for the whole chain; nothing but the weights ever crosses waves.
//   * The wave loads its 32 x 320 rows once, normalises them in registers (a row lives in the two lanes l, l + 32)
//     and keeps them as the 20 B-operand fragments of v_mfma_f32_32x32x16_bf16 (80 VGPRs).
//   * The hidden dimension is walked in chunks of 32: [h_c ; g_c]^T = W1_c X^T (two D^T accumulator tiles, K = 320),
//     hid_c = h_c * gelu(g_c) stays in the accumulator layout -- column (token) on the lane, hidden index in the
//     registers -- which IS the B operand of the second product out^T += W2_c hid_c^T (guide: "an accumulator tile as the
//     next MFMA's operand"); its permuted k order is baked into the packed W2.  out^T (320 x 32 f32) = 160 accumulator
//     registers.  With 1 wave per SIMD the kernel has the whole 512-register file.
//   * Weights: one linear stream per layer, already in LDS image order (XOR-swizzled 16-byte chunks), 60 KB per
//     iteration [W1 of chunk it | W2 of chunk it-2], copied by LDS-DMA into a 2-slot ring, one barrier per iteration
//     (the biases and the LayerNorm affine stay in LDS for the whole kernel).  Iteration `it` runs GEMM1(it), the GELU of
//     chunk it-1 and GEMM2(it-2): three independent register sets, so the GELU's VALU instructions and the ring's DMA
//     issues sit in the shadow of 60 MFMAs (<= 4 single-issue fillers per MFMA: MI355X_MICROARCH.md, "HIDDEN per gap").
//   * Epilogue: + b2 (accumulator init), D^T -> row-major through a wave-private LDS slab, + residual, 16-byte stores.
// The h16 GELU here is x * sigmoid(x (a + b u + c u^2)), u = min(x^2, 64): |err| <= 2.6e-5 absolute against the erf form
// (h16 resolution at 1.0 is 3.9e-3); the fp32 parity mode never takes this path.
#include "common.h"

namespace dsim {
namespace {

constexpr int RC = 320;                       // channels
constexpr int RKS = RC / 16;                  // 20 k-steps of 16
constexpr int RNB = RC / 32;                  // 10 output column blocks of 32
constexpr int RNCH = 4 * RC / 32;             // 40 hidden chunks of 32
constexpr int RITER = RNCH + 2;               // 42 ring iterations per row tile (even: slot parity == iteration parity)
constexpr int RW1B = 64 * RC * 2;             // 40960: [h rows 32 | g rows 32] x K 320, five swizzled [64][128 B] slabs
constexpr int RW2B = RC * 64;                 // 20480: [320 rows][32 hidden], swizzled 16-B chunks
constexpr int RCHB = RW1B + RW2B;             // 61440 bytes per ring slot / stream chunk
constexpr int RPIECES = RCHB / 1024;          // 60 LDS-DMA pieces: 15 per wave, no tail
constexpr int RPW = RPIECES / 4;
// resident vectors (f32): b1 [8C] GEGLU-interleaved, b2 [C], LayerNorm gamma / beta [C]
constexpr int RB1 = 0, RB2 = 8 * RC, RLG = 9 * RC, RLB = 10 * RC, RVEC = 11 * RC;
constexpr int RSCR = 32 * 144;                // per-wave transpose slab: 32 rows x (64 cols h16 + 16 pad)
constexpr int RLDS = 2 * RCHB + RVEC * 4 + 4 * RSCR;     // 155392
static_assert(RPIECES % 4 == 0 && RITER % 2 == 0, "ring geometry");

// stream chunk ci, 16-byte unit u: see the layout comment at the top
__global__ void pack_ff_stream_kernel(const h16* __restrict__ w1p, const h16* __restrict__ w2p, char* __restrict__ stream) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= RITER * (RCHB / 16)) return;
    const int ci = i / (RCHB / 16), o = (i - ci * (RCHB / 16)) * 16;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (o < RW1B) {
        if (ci < RNCH) {
            const int slab = o / 8192, row = (o % 8192) / 128, cpos = (o % 128) / 16;
            const int cl = cpos ^ ((row >> 1) & 7);
            v = *reinterpret_cast<const u32x4*>(w1p + (size_t)(64 * ci + row) * RC + 64 * slab + 8 * cl);
        }
    } else {
        if (ci >= 2) {
            const int c = ci - 2, o2 = o - RW1B;
            const int row = o2 / 64, cpos = (o2 % 64) / 16;
            const int cl = cpos ^ ((row >> 2) & 3);
            const int s = cl >> 1, hh = cl & 1;
            h16x8 e;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                e[j] = w2p[(size_t)row * (4 * RC) + 32 * c + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)];
            v = __builtin_bit_cast(u32x4, e);
        }
    }
    *reinterpret_cast<u32x4*>(stream + (size_t)ci * RCHB + o) = v;
}

struct FFParams {
    const h16* x;          // [M][320] residual stream (input of the LayerNorm and the residual)
    h16* out;              // [M][320]; may alias x
    const float* ln_g;
    const float* ln_b;
    const char* stream;     // RITER * RCHB bytes
    const float* b1;        // [8C] GEGLU-interleaved
    const float* b2;
    int M;
    float eps;
    unsigned x_bytes, stream_bytes;
    int stagger;            // s_nop 7 units (8 wait states each) wave w idles after every ring barrier, times w
};

// DBG (development builds only, tools/kbench ablations; the product instantiates DBG = 0): 1 no weight DMA, 2 no GELU,
// 4 no GEMM1 MFMAs, 8 no GEMM2 MFMAs, 16 no fragment reads
template <int DBG>
__global__ __launch_bounds__(256, 1) void ff_fused_kernel(const FFParams p, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr unsigned OOB = 0x80000000u;

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.stream, 0, (int)p.stream_bytes, 0x00020000);

    // ring: piece d (0..14) of this wave for stream chunk ci into the slot of parity sp; a wave copies 15 KB contiguous
    auto dma = [&](int sp, int ci, int d) {
        if (DBG & 1) return;
        char* dst = smem + sp * RCHB + wave * (RPW * 1024) + d * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (__attribute__((address_space(3))) void*)dst, 16, lane * 16,
                                                 ci * RCHB + wave * (RPW * 1024) + d * 1024, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < RPW; ++d) dma(0, 0, d);
    // the small vectors stay in LDS for the whole kernel
    float* const vec = reinterpret_cast<float*>(smem + 2 * RCHB);
    for (int i = tid; i < RVEC / 4; i += 256) {
        const int e = 4 * i;
        const float* src = e < RB2 ? p.b1 + e : (e < RLG ? p.b2 + (e - RB2) : (e < RLB ? p.ln_g + (e - RLG) : p.ln_b + (e - RLB)));
        *reinterpret_cast<f32x4*>(vec + e) = *reinterpret_cast<const f32x4*>(src);
    }

    // per-lane LDS read offsets inside a slot
    const int sw1 = (l31 >> 1) & 7;
    int w1off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w1off[kk] = l31 * 128 + (((2 * kk + half) ^ sw1) << 4);
    const int sw2 = (l31 >> 2) & 3;
    int w2off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) w2off[s] = RW1B + l31 * 64 + (((2 * s + half) ^ sw2) << 4);

    // rows of a tile as raw h16 B-operand fragments: lane (l31, half) holds columns 16 ks + 8 half .. + 7 of row l31
    u32x4 raw[RKS];
    auto load_rows = [&](int tile) {
        const int row = tile * 128 + wave * 32 + l31;
        const unsigned rbase = (tile < ntiles && row < p.M) ? (unsigned)row * (RC * 2) + half * 16 : OOB;
#pragma unroll
        for (int ks = 0; ks < RKS; ++ks) raw[ks] = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)(rbase + ks * 32), 0, 0);
    };
    load_rows(blockIdx.x);
    __syncthreads();      // the resident vectors are in LDS

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * 128 + wave * 32;
        // ---- LayerNorm in registers -> B-operand fragments ---------------------------------------
        // three passes over the h16 registers (sum, centred squares, normalise): an f32 copy of the rows would not fit
        // beside them in the 256 architectural VGPRs the VALU can address
        h16x8 X[RKS];
        {
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < RKS; ++ks) {
                const h16x8 t = __builtin_bit_cast(h16x8, raw[ks]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += (float)t[j];
            }
            {
                const unsigned u = __float_as_uint(sum);
                const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                sum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            const float mean = sum * (1.0f / RC);
            float sq = 0.f;
#pragma unroll
            for (int ks = 0; ks < RKS; ++ks) {
                asm volatile("" : "+v"(raw[ks]));      // (opaque: keeps hipcc from carrying the f32 conversions from pass to pass)
                const h16x8 t = __builtin_bit_cast(h16x8, raw[ks]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = (float)t[j] - mean; sq = fmaf(d, d, sq); }
            }
            {
                const unsigned u = __float_as_uint(sq);
                const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                sq = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            const float rstd = 1.0f / sqrtf(sq * (1.0f / RC) + p.eps);
#pragma unroll
            for (int ks = 0; ks < RKS; ++ks) {
                const int c = 16 * ks + 8 * half;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(vec + RLG + c), g1 = *reinterpret_cast<const f32x4*>(vec + RLG + c + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(vec + RLB + c), b1 = *reinterpret_cast<const f32x4*>(vec + RLB + c + 4);
                asm volatile("" : "+v"(raw[ks]));
                const h16x8 t = __builtin_bit_cast(h16x8, raw[ks]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    X[ks][j] = (h16)fmaf(((float)t[j] - mean) * rstd, g0[j], b0[j]);
                    X[ks][4 + j] = (h16)fmaf(((float)t[4 + j] - mean) * rstd, g1[j], b1[j]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- out^T accumulators start at b2 ----------------------------------------------------
        f32x16 out[RNB];
#pragma unroll
        for (int b = 0; b < RNB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(vec + RB2 + 32 * b + 8 * q + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) out[b][4 * q + e] = b4[e];
            }
        __builtin_amdgcn_sched_barrier(0);

        f32x16 aH[2], aG[2];      // GEMM1 accumulator sets (chunk parity)
        h16x8 wf[2][4];          // double-buffered weight fragments: group n reads set n & 1 while set (n + 1) & 1 loads
        h16x8 p0, p1;            // h * gelu(g) of chunk it-2: the two k-steps of GEMM2's B operand
        h16x8 q0, q1;            // ... of chunk it-1, being produced
        float hv[16];

        // wave w idles w * stagger units after each ring barrier: the loop lives inside one asm statement, so hipcc's
        // control-flow graph (and with it the pinned schedule and the register allocation) does not see a branch here
        auto stagger_wait = [&]() {
            int cnt = wave * p.stagger;
            asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lstag_end_%=\n.Lstag_%=:\n\ts_nop 7\n\ts_sub_u32 %0, %0, 1\n\t"
                         "s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 .Lstag_%=\n.Lstag_end_%=:" : "+s"(cnt) :: "scc");
        };
        // accumulators of chunk c start at its b1 slice
        auto init1 = [&](int c, f32x16& h, f32x16& gg) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bh = *reinterpret_cast<const f32x4*>(vec + RB1 + 64 * c + 8 * q + 4 * half);
                const f32x4 bg = *reinterpret_cast<const f32x4*>(vec + RB1 + 64 * c + 32 + 8 * q + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[4 * q + e] = bh[e]; gg[4 * q + e] = bg[e]; }
            }
        };
        // GEMM1 group i = k-steps 2i, 2i+1: fragments [h(2i), g(2i), h(2i+1), g(2i+1)]
        auto ld1 = [&](const char* slot, int i, h16x8 (&w)[4]) {
            if ((DBG & 16) && i > 0) return;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ks = 2 * i + u;
                const char* a = slot + (ks >> 2) * 8192 + w1off[ks & 3];
                w[2 * u] = *reinterpret_cast<const h16x8*>(a);
                w[2 * u + 1] = *reinterpret_cast<const h16x8*>(a + 32 * 128);
            }
        };
        auto mm1 = [&](int i, const h16x8 (&w)[4], f32x16& h, f32x16& gg) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (DBG & 4) { asm volatile("" :: "v"(w[2 * u]), "v"(w[2 * u + 1]), "v"(X[2 * i + u])); continue; }
                h = H16_MFMA_32x32x16(w[2 * u], X[2 * i + u], h, 0, 0, 0);
                gg = H16_MFMA_32x32x16(w[2 * u + 1], X[2 * i + u], gg, 0, 0, 0);
            }
        };
        // GEMM2 group j = output blocks 2j, 2j+1: fragments [b(2j) s0, b(2j) s1, b(2j+1) s0, b(2j+1) s1]
        auto ld2 = [&](const char* slot, int j, h16x8 (&w)[4]) {
            if (DBG & 16) return;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                w[2 * u] = *reinterpret_cast<const h16x8*>(slot + w2off[0] + (2 * j + u) * 2048);
                w[2 * u + 1] = *reinterpret_cast<const h16x8*>(slot + w2off[1] + (2 * j + u) * 2048);
            }
        };
        auto mm2 = [&](int j, const h16x8 (&w)[4]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (DBG & 8) { asm volatile("" :: "v"(w[2 * u]), "v"(w[2 * u + 1]), "v"(p0), "v"(p1)); continue; }
                out[2 * j + u] = H16_MFMA_32x32x16(w[2 * u], p0, out[2 * j + u], 0, 0, 0);
                out[2 * j + u] = H16_MFMA_32x32x16(w[2 * u + 1], p1, out[2 * j + u], 0, 0, 0);
            }
        };
        // element e of h * gelu(g) of the previous chunk; pairs are packed as soon as the odd one exists
        auto geglu1 = [&](int e, const f32x16& h, const f32x16& gg) {
            hv[e] = (DBG & 2) ? h[e] + gg[e] : h[e] * gelu_fast(gg[e]);
            if (e & 1) {
                if (e < 8) { q0[e - 1] = (h16)hv[e - 1]; q0[e] = (h16)hv[e]; }
                else { q1[e - 9] = (h16)hv[e - 1]; q1[e - 8] = (h16)hv[e]; }
            }
        };
        // One iteration, pinned group by group (left alone hipcc issues every ds_read right before its MFMA, and a lone wave
        // per SIMD then waits out the LDS latency 60 times per iteration): the fragments of group n+1 are requested, then
        // group n's four MFMAs issue with their share of the GELU and of the ring's DMA pieces between them.  Groups 0-9:
        // GEMM1(c); groups 10-14: GEMM2(c-2); the 16 GELU elements of chunk c-1 and the 15 DMA pieces of the next iteration
        // (those in the first half of the groups, so that they have landed by the next barrier) are dealt over the groups.
        // PAR = parity of the iteration (ring slot and GEMM1 accumulator set), a literal; C = chunk index (may be a variable).
#define RR_SB __builtin_amdgcn_sched_barrier(0)
#define RR_MIX(NV)                                                                                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
#define RR_BODY(PAR, C, G1, GL, G2, NEXT_CI)                                                                          \
        {                                                                                                             \
            constexpr int NG = ((G1) ? 10 : 0) + ((G2) ? 5 : 0), NH = NG == 15 ? 12 : (NG == 10 ? 8 : 4);             \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
            __syncthreads();                                                                                          \
            stagger_wait();                        /* de-phase the four waves */                                      \
            const char* slot = smem + (PAR) * RCHB;                                                                   \
            if (G1) { init1((C), aH[PAR], aG[PAR]); ld1(slot, 0, wf[0]); } else ld2(slot, 0, wf[0]);                  \
            RR_SB;                                                                                                    \
            _Pragma("unroll") for (int n = 0; n < NG; ++n) {                                                          \
                const int gi = (G1) ? n : n + 10;          /* group id: 0-9 GEMM1, 10-14 GEMM2 */                     \
                if (n + 1 < NG) { if (gi + 1 < 10) ld1(slot, gi + 1, wf[(n + 1) & 1]); else ld2(slot, gi + 1 - 10, wf[(n + 1) & 1]); } \
                RR_SB;                                                                                                \
                if (gi < 10) mm1(gi, wf[n & 1], aH[PAR], aG[PAR]); else mm2(gi - 10, wf[n & 1]);                      \
                _Pragma("unroll") for (int d = 0; d < RPW; ++d) if ((d * NH) / RPW == n) dma((PAR) ^ 1, (NEXT_CI), d); \
                if (GL) { _Pragma("unroll") for (int e = 0; e < 16; ++e) if ((e * NG) / 16 == n) geglu1(e, aH[(PAR) ^ 1], aG[(PAR) ^ 1]); } \
                RR_MIX(4)                                                                                             \
                RR_SB;                                                                                                \
            }                                                                                                         \
            if (GL) { p0 = q0; p1 = q1; }                                                                             \
        }
        RR_BODY(0, 0, true, false, false, 1)
        RR_BODY(1, 1, true, true, false, 2)
        for (int c = 2; c < RNCH; c += 2) {          // two iterations per trip: every register-set index stays static
            RR_BODY(0, c, true, true, true, c + 1)
            RR_BODY(1, c + 1, true, true, true, c + 2)
        }
        // X is dead from here: request the next tile's rows under the two drain iterations
        load_rows(tile + (int)gridDim.x);
        RR_BODY(0, RNCH, false, true, true, RNCH + 1)
        RR_BODY(1, RNCH + 1, false, false, true, 0)
#undef RR_BODY
#undef RR_MIX
#undef RR_SB

        // ---- epilogue: D^T -> row-major through the wave's LDS slab, + residual, 16-byte stores ----
        char* const slab = smem + 2 * RCHB + RVEC * 4 + wave * RSCR;
        // read-back geometry: per pass (64 columns) 32 rows x 8 chunks of 16 B = 256 pieces, 4 per lane
        u32x4 res[RNB / 2][4];
#pragma unroll
        for (int pp = 0; pp < RNB / 2; ++pp)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = lane + it * 64, r = idx >> 3, c = idx & 7;
                const unsigned go = (m0 + r) < p.M ? (unsigned)(m0 + r) * (RC * 2) + (unsigned)(pp * 128 + c * 16) : OOB;
                res[pp][it] = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)go, 0, 0);
            }
#pragma unroll
        for (int pp = 0; pp < RNB / 2; ++pp) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (h16)out[2 * pp + bb][4 * q + e];
                    *reinterpret_cast<h16x4*>(slab + l31 * 144 + (bb * 32 + 8 * q + 4 * half) * 2) = pk;
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = lane + it * 64, r = idx >> 3, c = idx & 7;
                const unsigned go = (m0 + r) < p.M ? (unsigned)(m0 + r) * (RC * 2) + (unsigned)(pp * 128 + c * 16) : OOB;
                const h16x8 t = *reinterpret_cast<const h16x8*>(slab + r * 144 + c * 16);
                const h16x8 rr = __builtin_bit_cast(h16x8, res[pp][it]);
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (h16)((float)t[e] + (float)rr[e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rO, (int)go, 0, 0);
            }
        }
    }
    // the last DMA pieces of this workgroup (the next tile's chunk 0) are still in flight: drain before exit
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- row-resident Linear (optionally behind a LayerNorm) for K = 320 -----------------------------------------------------
// out[M][N] = LN?(x)[M][320] W^T (+ bias), N a multiple of 64: the fused to_q/to_k/to_v projection behind norm1 (N = 960) and
// attn2.to_q behind norm2 (hacked_modules.py:88-116).  The tiled GEMM is latency-bound on these: a K of five tiles leaves its
// two-stage ring one 1.2 us K step of cover for activations that stream from HBM exactly once (726 TF/s, 3.0 TB/s at N = 960),
// and the LayerNorm in front costs a full read + write of the tensor.  Here a wave keeps its 32 rows in registers
// (normalised on arrival), the weights stream through a 2-slot LDS ring in 32-column blocks, the output leaves in
// 128-byte row segments per pair of blocks -- HBM-bound, one pass.  Two 4-wave workgroups per CU (<= 256 registers, 66 KB
// of LDS): one workgroup's row loads and stores sit under the other's MFMAs.
constexpr int LCHB = 32 * RC * 2;              // 20480 bytes per ring slot: W rows [32 b, 32 b + 32) x K 320, five [32][128 B] slabs
constexpr int LPW = LCHB / 1024 / 4;           // 5 DMA pieces per wave per block
constexpr int LVEC = 2 * RC;                   // resident f32: LayerNorm gamma, beta
constexpr int LSCR = 16 * 144;                 // per-wave transpose slab: 16 rows x (64 cols h16 + 16 pad)
constexpr int LLDS = 2 * LCHB + LVEC * 4 + 4 * LSCR;      // 52736: three workgroups per CU

__global__ void pack_rowlin_stream_kernel(const h16* __restrict__ wp, char* __restrict__ stream, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N / 32 * (LCHB / 16)) return;
    const int b = i / (LCHB / 16), o = (i - b * (LCHB / 16)) * 16;
    const int slab = o / 4096, row = (o % 4096) / 128, cpos = (o % 128) / 16;
    const int cl = cpos ^ ((row >> 1) & 7);
    *reinterpret_cast<u32x4*>(stream + (size_t)b * LCHB + o) =
        *reinterpret_cast<const u32x4*>(wp + (size_t)(32 * b + row) * RC + 64 * slab + 8 * cl);
}

struct RLParams {
    const h16* x;
    h16* out;
    const float* ln_g;      // null: no LayerNorm
    const float* ln_b;
    const char* stream;
    int M, N;
    float eps;
    unsigned x_bytes, out_bytes, stream_bytes;
};

template <int DBG>      // DBG: kbench ablation masks (1 no output stores, 2 no MFMAs, 4 no row loads, 8 no fragment reads); 0 in the product
__global__ __launch_bounds__(256, 3) void rowlin_kernel(const RLParams p, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.stream, 0, (int)p.stream_bytes, 0x00020000);
    const int nblk = p.N / 32;                 // even
    auto dma = [&](int sp, int blk) {           // this wave's five pieces of weight block `blk` into the slot of parity sp
        char* dst = smem + sp * LCHB + wave * (LPW * 1024);
#pragma unroll
        for (int d = 0; d < LPW; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, lane * 16,
                                                     blk * LCHB + wave * (LPW * 1024) + d * 1024, 0, 0);
    };
    dma(0, 0);
    float* const vec = reinterpret_cast<float*>(smem + 2 * LCHB);         // [gamma 320 | beta 320]
    if (p.ln_g)
        for (int i = tid; i < LVEC; i += 256) vec[i] = i < RC ? p.ln_g[i] : p.ln_b[i - RC];
    char* const slab = smem + 2 * LCHB + LVEC * 4 + wave * LSCR;
    const int sw1 = (l31 >> 1) & 7;
    int woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) woff[kk] = l31 * 128 + (((2 * kk + half) ^ sw1) << 4);
    __syncthreads();
    int blk = 0;                                // weight block the NEXT ring step consumes (wraps at nblk: same weights for every tile)

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * 128 + wave * 32;
        // the rows live in ONE register array: raw h16 on arrival, overwritten in place by their normalised values
        u32x4 xr[RKS];
        {
            const int row = m0 + l31;
            const unsigned rbase = row < p.M ? (unsigned)row * (RC * 2) + half * 16 : OOB;
#pragma unroll
            for (int ks = 0; ks < RKS; ++ks)
                xr[ks] = (DBG & 4) ? u32x4{(unsigned)ks, 1u, 2u, rbase} : __builtin_amdgcn_raw_buffer_load_b128(rX, (int)(rbase + ks * 32), 0, 0);
            if (p.ln_g) {                       // the LayerNorm of ff_fused_kernel (three passes over the h16 registers)
                float sum = 0.f;
#pragma unroll
                for (int ks = 0; ks < RKS; ++ks) {
                    const h16x8 t = __builtin_bit_cast(h16x8, xr[ks]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += (float)t[j];
                }
                {
                    const unsigned u = __float_as_uint(sum);
                    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    sum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
                const float mean = sum * (1.0f / RC);
                float sq = 0.f;
#pragma unroll
                for (int ks = 0; ks < RKS; ++ks) {
                    asm volatile("" : "+v"(xr[ks]));
                    const h16x8 t = __builtin_bit_cast(h16x8, xr[ks]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = (float)t[j] - mean; sq = fmaf(d, d, sq); }
                }
                {
                    const unsigned u = __float_as_uint(sq);
                    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    sq = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
                const float rstd = 1.0f / sqrtf(sq * (1.0f / RC) + p.eps);
#pragma unroll
                for (int ks = 0; ks < RKS; ++ks) {
                    const int c = 16 * ks + 8 * half;
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(vec + c), g1 = *reinterpret_cast<const f32x4*>(vec + c + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(vec + RC + c), b1 = *reinterpret_cast<const f32x4*>(vec + RC + c + 4);
                    asm volatile("" : "+v"(xr[ks]));
                    const h16x8 t = __builtin_bit_cast(h16x8, xr[ks]);
                    h16x8 y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        y[j] = (h16)fmaf(((float)t[j] - mean) * rstd, g0[j], b0[j]);
                        y[4 + j] = (h16)fmaf(((float)t[4 + j] - mean) * rstd, g1[j], b1[j]);
                    }
                    xr[ks] = __builtin_bit_cast(u32x4, y);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // one ring step: the 20 MFMAs of weight block `blk` (32 output columns) into `a`, fragments double-buffered by 64-wide K slab
        h16x8 wf[2][2];
        auto step = [&](int sp, f32x16& a) {
            // the slot's five DMA pieces must have landed; the four row-segment stores of the previous pair were issued after the
            // pieces of an even step (vmcnt retires in issue order on gfx9): leave them in flight there
            if (sp == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int nb = blk + 1 == nblk ? 0 : blk + 1;
            dma(sp ^ 1, nb);
            blk = nb;
            const char* slot = smem + sp * LCHB;
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.f;
            auto ld = [&](int i, h16x8 (&w)[2]) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ks = 2 * i + u;
                    w[u] = *reinterpret_cast<const h16x8*>(slot + (ks >> 2) * 4096 + woff[ks & 3]);
                }
            };
            ld(0, wf[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                if (i < 9 && !(DBG & 8)) ld(i + 1, wf[(i + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (!(DBG & 2)) a = H16_MFMA_32x32x16(wf[i & 1][u], __builtin_bit_cast(h16x8, xr[2 * i + u]), a, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // a pair of blocks = 64 output columns = 128-byte row segments: D^T -> row-major through the wave's 16-row slab in two
        // passes (rows 0-15, 16-31), 16-byte stores of full lines
        auto store_pair = [&](const f32x16& a0, const f32x16& a1, int col0) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                asm volatile("" ::: "memory");       // pass 1 overwrites the slab pass 0 is read from
                if ((l31 >> 4) == ph) {
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            h16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = (h16)(bb ? a1[4 * q + e] : a0[4 * q + e]);
                            *reinterpret_cast<u32x2*>(slab + (l31 & 15) * 144 + (bb * 32 + 8 * q + 4 * half) * 2) = __builtin_bit_cast(u32x2, pk);
                        }
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int idx = lane + it * 64, r = idx >> 3, c = idx & 7, row = m0 + 16 * ph + r;
                    const unsigned go = row < p.M ? ((unsigned)row * (unsigned)p.N + (unsigned)col0) * 2u + (unsigned)c * 16u : OOB;
                    const u32x4 t = *reinterpret_cast<const u32x4*>(slab + r * 144 + c * 16);
                    if (!(DBG & 1) || t[0] == 0x12345u) __builtin_amdgcn_raw_buffer_store_b128(t, rO, (int)go, 0, 0);
                }
            }
        };
        f32x16 a0, a1;
        for (int pr = 0; pr < nblk / 2; ++pr) {
            step(0, a0);
            step(1, a1);
            store_pair(a0, a1, 64 * pr);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

#ifdef DSIM_DEVTOOLS
int g_rl_dbg = 0;
int g_ff_dbg = 0;
int g_ff_stagger = -1;     // -1 = the product default
int g_rl_wpc = 3;          // rowlin_kernel's persistent workgroups per CU
#endif

size_t ff_stream_bytes(int C) { return C == RC ? (size_t)RITER * RCHB : 0; }

int pack_ff_stream(const void* w1_packed, const void* w2_packed, void* stream, int C, hipStream_t s) {
    if (C != RC) return DSIM_ERR_INVALID;
    const int n = RITER * (RCHB / 16);
    hipLaunchKernelGGL(pack_ff_stream_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const h16*)w1_packed,
                       (const h16*)w2_packed, (char*)stream);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

int launch_ff_fused(const FFArgs& a, hipStream_t s) {
#ifdef DSIM_HAS_F16_TWINS
    if (a.dtype == DSIM_F16) return launch_ff_fused_f16(a, s);
#endif
    if (a.dtype != DSIM_H16) return DSIM_ERR_INVALID;
    if (a.C != RC || a.M < 1 || (size_t)a.M * RC * 2 >= 0x7fffffffull) return DSIM_ERR_INVALID;
    FFParams p;
    p.x = (const h16*)a.x; p.out = (h16*)a.out; p.ln_g = a.ln_g; p.ln_b = a.ln_b; p.stream = (const char*)a.stream;
    p.b1 = a.b1; p.b2 = a.b2; p.M = a.M; p.eps = a.eps;
    p.x_bytes = (unsigned)((size_t)a.M * RC * 2);
    p.stream_bytes = (unsigned)((size_t)RITER * RCHB);
    // The four waves of a workgroup leave every ring barrier together and issue their DMA pieces and fragment reads at the same
    // instants; de-phased by one 8-wait-state unit per wave index the kernel runs 1.7 % faster (interleaved-round sweep in
    // profiles/r03_ff_fused_ablation.txt: 0 -> 1.208, 1 -> 1.187, 2 -> 1.218, 3 -> 1.246 ms).
    p.stagger = 1;
#ifdef DSIM_DEVTOOLS
    if (g_ff_stagger >= 0) p.stagger = g_ff_stagger;
#endif
    const int ntiles = (a.M + 127) / 128;
    const int grid = ntiles < cu_count() ? ntiles : cu_count();
#ifdef DSIM_DEVTOOLS
    switch (g_ff_dbg) {
#define X(d) case d: { static DeviceOnce o; auto k = ff_fused_kernel<d>; CK_ONCE(o, k, RLDS); hipLaunchKernelGGL(k, dim3(grid), dim3(256), RLDS, s, p, ntiles); DSIM_HIP_CHECK(hipGetLastError()); return DSIM_OK; }
        X(1) X(2) X(3) X(12) X(13) X(15) X(16) X(31)
#undef X
        default: break;
    }
#endif
    static DeviceOnce once;
    auto kern = ff_fused_kernel<0>;
    CK_ONCE(once, kern, RLDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), RLDS, s, p, ntiles);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

size_t rowlin_stream_bytes(int C, int N) { return (C == RC && N % 64 == 0 && N <= 960) ? (size_t)N / 32 * LCHB : 0; }

int pack_rowlin_stream(const void* w_packed, void* stream, int C, int N, hipStream_t s) {
    if (!rowlin_stream_bytes(C, N)) return DSIM_ERR_INVALID;
    const int n = N / 32 * (LCHB / 16);
    hipLaunchKernelGGL(pack_rowlin_stream_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const h16*)w_packed, (char*)stream, N);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

int launch_rowlin(const RowLinArgs& a, hipStream_t s) {
#ifdef DSIM_HAS_F16_TWINS
    if (a.dtype == DSIM_F16) return launch_rowlin_f16(a, s);
#endif
    if (a.dtype != DSIM_H16) return DSIM_ERR_INVALID;
    if (!rowlin_stream_bytes(a.C, a.N) || a.M < 1 || (size_t)a.M * a.N * 2 >= 0x7fffffffull || !a.ln_g != !a.ln_b) return DSIM_ERR_INVALID;
    RLParams p;
    p.x = (const h16*)a.x; p.out = (h16*)a.out; p.ln_g = a.ln_g; p.ln_b = a.ln_b; p.stream = (const char*)a.stream;
    p.M = a.M; p.N = a.N; p.eps = a.eps;
    p.x_bytes = (unsigned)((size_t)a.M * RC * 2); p.out_bytes = (unsigned)((size_t)a.M * a.N * 2);
    p.stream_bytes = (unsigned)rowlin_stream_bytes(a.C, a.N);
    const int ntiles = (a.M + 127) / 128;
#ifdef DSIM_DEVTOOLS
    const int wpc = g_rl_wpc;        // kbench occupancy probe: workgroups per CU
#else
    constexpr int wpc = 3;
#endif
    const int grid = ntiles < wpc * cu_count() ? ntiles : wpc * cu_count();
#ifdef DSIM_DEVTOOLS
    switch (g_rl_dbg) {
#define X(d) case d: { static DeviceOnce o; auto k = rowlin_kernel<d>; CK_ONCE(o, k, LLDS); hipLaunchKernelGGL(k, dim3(grid), dim3(256), LLDS, s, p, ntiles); DSIM_HIP_CHECK(hipGetLastError()); return DSIM_OK; }
        X(1) X(2) X(4) X(8) X(10) X(15)
#undef X
        default: break;
    }
#endif
    static DeviceOnce once;
    auto kern = rowlin_kernel<0>;
    CK_ONCE(once, kern, LLDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LLDS, s, p, ntiles);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

}  // namespace dsim
