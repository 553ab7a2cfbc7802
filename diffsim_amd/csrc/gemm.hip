// Implicit-GEMM MFMA kernel for gfx950: linear / 1x1 conv / 3x3 conv (stride 1|2, folded
// nearest-2x upsample, channel-concat of two sources) with fused bias / residual / GEGLU
// epilogues.  One template serves the bf16 production path (v_mfma_f32_32x32x16_bf16) and the
// fp32 parity path (v_mfma_f32_32x32x2_f32, an exact f32 fma chain).
//
// Replaces what the reference dispatches to cuDNN/cuBLAS through diffusers' ResnetBlock2D /
// Transformer2DModel / Attention / FeedForward modules (control flow
// in /root/reference/diffsim/hacked_modules.py:17-136, 261-434).
//
// Data layout: activations token-major [M][C]; weights packed [N][K] with K contiguous
// (conv: k = tap*Cin + c).  LDS tiles are [rows][128 B] with the 16-B chunk index XOR-swizzled
// by (row>>1)&7 so that ds_read_b128 fragment reads are bank-conflict free; tiles are filled by
// LDS-DMA (global_load_lds_dwordx4), whose lane-linear destination means the swizzle is applied
// to the per-lane SOURCE address.  Zero padding of conv borders and ragged M/N edges comes from
// pointing those lanes at a 16-byte zero page.
#include "common.h"

namespace dsim {

// tile choice: 160-wide tiles when N allows (every SD channel count is a multiple of 160),
// 256-row tiles once they still give >= one workgroup per CU.
int g_force_bm = 0;
// Tile choice.  Small problems: 128-row tiles, 4 waves, two workgroups per CU (160-wide when N
// allows -- every SD channel count is a multiple of 160 -- else 128).  bf16 problems with enough
// 256-row tiles to fill the chip: 256 x 320 (or 256 x 256) tiles, 8 waves as 4 x 2.
void gemm_tile_choice(const GemmArgs& a, int* bm, int* bn) {
    const bool geglu = a.epi == EPI_GEGLU;
    const bool n320 = !geglu && a.N % 320 == 0, n256 = a.N % 256 == 0;
    int want256 = 0;
    if (n320 || n256) {
        const int bnb = n320 ? 320 : 256;
        const long tiles = (long)((a.M + 255) / 256) * (a.N / bnb);
        want256 = tiles >= 256;
    }
    if (g_force_bm == 128) want256 = 0;                       // development override (kbench A/B)
    if (g_force_bm == 256) want256 = (n320 || n256);
    if (want256) { *bm = 256; *bn = n320 ? 320 : 256; return; }
    *bm = 128;
    *bn = (a.N % 160 == 0 && !geglu) ? 160 : 128;
}

namespace {

template <typename T> struct Traits;
template <> struct Traits<bf16> {
    static constexpr int BK = 64;     // elements per 128-byte LDS row
    static constexpr int KSUB = 4;    // 16-deep MFMA steps per K tile
    static constexpr int VEC = 8;     // elements per 16-byte chunk
};
template <> struct Traits<float> {
    static constexpr int BK = 32;
    static constexpr int KSUB = 2;
    static constexpr int VEC = 4;
};

template <typename T> struct Vec16T;
template <> struct Vec16T<bf16> { typedef bf16x8 type; };
template <> struct Vec16T<float> { typedef f32x4 type; };

struct FragF32 { f32x4 lo, hi; };

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void load_frag(bf16x8& f, const char* row, int c0, int sw) {
    f = *reinterpret_cast<const bf16x8*>(row + ((c0 ^ sw) << 4));
}
__device__ __forceinline__ void load_frag(FragF32& f, const char* row, int c0, int sw) {
    f.lo = *reinterpret_cast<const f32x4*>(row + ((c0 ^ sw) << 4));
    f.hi = *reinterpret_cast<const f32x4*>(row + (((c0 + 1) ^ sw) << 4));
}
__device__ __forceinline__ void mma(const bf16x8& a, const bf16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// f32: the 16-deep step is 8 exact-f32 MFMAs of depth 2; lane half h supplies k = 8h + j to
// MFMA j for both operands, so the k pairing is consistent (any k order is a valid dot product).
__device__ __forceinline__ void mma(const FragF32& a, const FragF32& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], c, 0, 0, 0);
}
template <typename T> struct FragOf { typedef bf16x8 type; };
template <> struct FragOf<float> { typedef FragF32 type; };

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. f32 rounding level): one v_rcp, one
// v_exp and 7 FMAs instead of libm erff's ~30-instruction branchy polynomial -- the GEGLU epilogue
// evaluates it 64x per thread per tile and was VALU-bound on it.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    pl *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = fmaf(-pl, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

// LDS: two staging buffers for the K loop; the epilogue's four wave-private transpose slabs reuse them
template <typename T, int BM, int BN, bool GEGLU, int WM, int WN, int NST>
constexpr int gemm_lds_bytes() {
    const int stages = NST * (BM + BN) * 128;
    const int epi = WM * WN * 32 * ((GEGLU ? BN / WN / 2 : BN / WN) * (int)sizeof(T) + 16);
    return stages > epi ? stages : epi;
}

// WM x WN waves; each owns a (BM/WM) x (BN/WN) sub-tile, so an A fragment is reused by BN/WN/32 MFMAs and
// a B fragment by BM/WM/32: with 64 x 160 per wave the LDS read traffic per MFMA is 0.7 fragments
// against 1.2 for 32 x 160 (the 4x1 layout) -- the LDS port, not the MFMA pipe, was the limiter there.
// NST LDS stages:
//   NST == 2: stage t+1 is issued at the top of step t and drained (vmcnt(0)) at its end;
//   NST == 3: stage t+2 is issued at the top of step t and only stage t+1 -- a counted vmcnt --
//             must have landed before the step's barrier, so a whole K tile of loads stays in
//             flight across every barrier (raw s_barrier: __syncthreads() would drain the DMA).
template <typename T, int BM, int BN, int MODE, bool GEGLU, int WM, int WN, int NST>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_kernel(const GemmArgs p, const int tilesN) {
    constexpr int NW = WM * WN;
    constexpr int BK = Traits<T>::BK;
    constexpr int KSUB = Traits<T>::KSUB;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int WBN = BN / WN;                           // columns per wave
    constexpr int NA = BM / (NW * 8);                      // 8-row DMA pieces of A per wave per stage
    constexpr int NBP = BN / 8;                            // 8-row DMA pieces of B per stage (all waves)
    constexpr int NB = (NBP + NW - 1) / NW;                // ... per wave (the last one may be partial)
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    typedef typename FragOf<T>::type Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware bijective remap: the 8 XCDs take consecutive block ids round-robin; give each
    // XCD a contiguous run of logical tiles so that tiles sharing an A panel share an L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = (bid / tilesN) * BM, n0 = (bid % tilesN) * BN;

    const int lrow = lane >> 3;                                   // row inside an 8-row DMA piece
    const int wrow = wave * 8 + lrow;                             // row inside a 32-row group
    const unsigned celb = ((lane & 7) ^ ((wrow >> 1) & 7)) * 16;  // swizzled source chunk (bytes)
    constexpr unsigned OOB = 0x80000000u;                         // voffset beyond every buffer -> DMA writes zeros

    // Buffer descriptors: every per-lane part of an address lives in a 32-bit voffset, the K
    // advance in a scalar soffset, so the steady-state staging costs no VALU at all, and rows that
    // must read as zero (conv padding, ragged M/N edges) simply carry an out-of-range voffset.
    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A0, 0, (int)p.a0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A1 ? p.A1 : p.A0), 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);

    // ---- per-thread A row state ----------------------------------------------------------
    int a_iy0[NA], a_ix0[NA];
    unsigned a_base[NA], a_voff[NA];
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + i * (NW * 8) + wrow;
        if (MODE == GEMM_CONV3) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_iy0[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
            a_ix0[i] = ox * p.stride - p.pad;
            a_base[i] = (unsigned)b * (unsigned)(p.Hin * p.Win);
            a_voff[i] = OOB;
        } else {
            a_iy0[i] = a_ix0[i] = 0;
            a_base[i] = (m < p.M) ? (unsigned)m : OOB;
            a_voff[i] = (m < p.M) ? (unsigned)m * (unsigned)p.C0 * (unsigned)sizeof(T) + celb : OOB;
        }
    }
    unsigned b_voff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + i * (NW * 8) + wrow;
        b_voff[i] = (n < p.N) ? (unsigned)n * (unsigned)p.K * (unsigned)sizeof(T) + celb : OOB;
    }
    // B pieces this wave really issues per stage (wave-uniform): the tail piece exists only for
    // the first NBP % NW waves
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool b_tail = (NBP % NW == 0) || wave_u < (NBP % NW);

    auto stage = [&](int t, int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
        const int k0 = t * BK;
        int soff;
        bool second = false;
        if (MODE == GEMM_CONV3) {
            const int tap = k0 / p.C0;
            const int koff = k0 - tap * p.C0;
            soff = koff * (int)sizeof(T);
            if (koff == 0) {   // first K tile of a tap: re-derive the gathered pixel of every row
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                    const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                    const unsigned pix = a_base[i] + (unsigned)((iy >> p.ups) * p.Win + (ix >> p.ups));
                    a_voff[i] = ok ? pix * (unsigned)p.C0 * (unsigned)sizeof(T) + celb : OOB;
                }
            }
        } else {
            second = k0 >= p.C0;
            soff = (second ? k0 - p.C0 : k0) * (int)sizeof(T);
            if (second && k0 == p.C0) {   // first K tile of the concatenated second source
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    a_voff[i] = a_base[i] != OOB ? a_base[i] * (unsigned)p.C1 * (unsigned)sizeof(T) + celb : OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            auto lds = (__attribute__((address_space(3))) void*)(sa + (i * NW + wave) * 1024);
            if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, lds, 16, (int)a_voff[i], soff, 0, 0);
            else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, lds, 16, (int)a_voff[i], soff, 0, 0);
        }
        const int soffw = k0 * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (i + 1 < NB || b_tail)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(sb + (i * NW + wave) * 1024),
                                                         16, (int)b_voff[i], soffw, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.K / BK;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int half = lane >> 5;
    const int sw = (lane >> 1) & 7;                  // == (row>>1)&7 for row = 32*x + (lane&31)
    const int frow = (lane & 31) * 128;

    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE + wm * (BM / WM) * 128 + frow;
        const char* sb = smem + buf * STAGE + A_BYTES + wn * WBN * 128 + frow;
        // fragments are register double-buffered: the ds_reads of step kk+1 are in flight while the
        // MFMAs of step kk run, so only the first read of a K tile exposes LDS latency
        Frag fa[2][TM], fb[2][TN];
        auto load_set = [&](int kk, int set) {
            const int c0 = (sizeof(T) == 2) ? (2 * kk + half) : (4 * kk + 2 * half);
#pragma unroll
            for (int i = 0; i < TM; ++i) load_frag(fa[set][i], sa + i * 32 * 128, c0, sw);
#pragma unroll
            for (int j = 0; j < TN; ++j) load_frag(fb[set][j], sb + j * 32 * 128, c0, sw);
        };
        load_set(0, 0);
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk) {
            if (kk + 1 < KSUB) load_set(kk + 1, (kk + 1) & 1);
            // pin the order "all reads of step kk+1, then all MFMAs of step kk": left alone, hipcc
            // sinks each ds_read to just before its MFMA and every MFMA then waits out LDS latency
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    mma(fb[kk & 1][j], fa[kk & 1][i], acc[i][j]);   // D^T: rows = n (registers), cols = m (lane)
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (NST == 2) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk) stage(t + 1, cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // loads of one stage by this wave: NA + NB (or NB-1 without the tail piece)
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        int buf = 0;
        for (int t = 0; t < nk; ++t) {
            // stage t must have landed; the newer stage t+1 (if any) may stay in flight
            if (t + 1 < nk) {
                if (b_tail) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
                else        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB - 1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();       // every wave's share of stage t is in LDS; step t-1 fully consumed
            if (t + 2 < nk) stage(t + 2, buf == 0 ? 2 : buf - 1);      // into the buffer step t-1 read
            compute(buf);
            buf = buf == 2 ? 0 : buf + 1;
        }
        __builtin_amdgcn_s_barrier();           // all waves done with the last stage before the epilogue reuses LDS
    }

    // ---- epilogue -------------------------------------------------------------------------
    // The accumulators hold D^T: lane = output row m (lane&31), registers = 16 output columns
    // n = 8*(r>>2) + 4*half + (r&3) of a 32-wide block.  Each wave transposes its 32-row slab
    // through a private LDS region (the K loop's last barrier freed the staging buffers), then
    // streams it out row-contiguously: 16-byte coalesced residual loads and stores.
    constexpr int ES = sizeof(T);
    constexpr int OUTW = GEGLU ? WBN / 2 : WBN;        // output columns this wave produces
    constexpr int RSO = OUTW * ES + 16;                 // staging row stride (bytes)
    constexpr int CPR = OUTW * ES / 16;                 // 16-byte chunks per output row
    char* const wst = smem + wave * (32 * RSO);
    T* const out = (T*)p.out;
    const T* const res = (const T*)p.residual;
    const int nw0 = n0 + wn * WBN;                     // first packed weight row of this wave
    const int nout0 = GEGLU ? (nw0 >> 1) : nw0;
    const int Nout = GEGLU ? (p.N >> 1) : p.N;
    const int l31 = lane & 31;
    typedef typename Vec16T<T>::type V16;
    constexpr int NIT = (32 * CPR + 63) / 64;          // read-back iterations per 32-row slab
    const bool has_res = p.epi == EPI_RESIDUAL;
    const bool slow = p.act != 0 || p.gate != nullptr;  // DiT epilogues: activation / adaLN gate after the bias
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow0 = m0 + wm * (BM / WM) + i * 32;
        // residual prefetch: every 16-byte piece this lane will add is requested before the slab is
        // transposed, so the HBM latency hides under the register phase instead of serialising the stores
        V16 rres[NIT];
        if (has_res) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = lane + it * 64;
                const int row = idx / CPR, c = idx - row * CPR;
                const int m = mrow0 + row, ncol = c * VEC;
                if (idx < 32 * CPR && m < p.M && nout0 + ncol < Nout)
                    rres[it] = *reinterpret_cast<const V16*>(res + (size_t)m * p.ldo + nout0 + ncol);
            }
        }
        // register phase: bias (f32, before the one rounding to T) and the D^T -> row-major transpose through LDS
        const bool odd_half = p.bias2 && (((mrow0 + l31) / p.rows_per_batch) & 1);
#pragma unroll
        for (int j = 0; j < TN; j += (GEGLU ? 2 : 1)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
                if (GEGLU) {
                    // packed weight rows alternate 32-row blocks [h-block, g-block]
                    const int nh = nw0 + j * 32 + 8 * g + 4 * half;
                    f32x4 bh = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias && nh < p.N) {
                        bh = *reinterpret_cast<const f32x4*>(p.bias + nh);
                        bg = *reinterpret_cast<const f32x4*>(p.bias + nh + 32);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = (acc[i][j][4 * g + e] + bh[e]) * gelu_erf(acc[i][j + 1][4 * g + e] + bg[e]);
                } else {
                    const int nb = nw0 + j * 32 + 8 * g + 4 * half;
                    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias && nb < p.N) {
                        b4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
                        if (p.bias2) {
                            const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.bias2 + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) b4[e] = odd_half ? c4[e] : b4[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + b4[e];
                }
                const int col = (GEGLU ? (j >> 1) : j) * 32 + 8 * g + 4 * half;
                char* dst = wst + l31 * RSO + col * ES;
                if constexpr (sizeof(T) == 2) {
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (bf16)v[e];
                    *reinterpret_cast<bf16x4*>(dst) = pk;
                } else {
                    f32x4 pk = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(dst) = pk;
                }
            }
        }
        // read-back phase (same wave: LDS operations of one wave execute in order)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + it * 64;
            const int row = idx / CPR, c = idx - row * CPR;
            const int m = mrow0 + row, ncol = c * VEC;
            if (idx < 32 * CPR && m < p.M && nout0 + ncol < Nout) {
                const size_t o = (size_t)m * p.ldo + nout0 + ncol;
                const V16 t = *reinterpret_cast<const V16*>(wst + row * RSO + c * 16);
                if (!has_res && !slow) {                 // plain projection: LDS -> HBM copy
                    *reinterpret_cast<V16*>(out + o) = t;
                    continue;
                }
                float v[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = (float)t[e];
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {      // tanh-GELU: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
                        const float x = v[e], u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
                        const float ex = __builtin_amdgcn_exp2f(2.8853900817779268f * u);      // e^(2u)
                        v[e] = 0.5f * x * (1.0f + (1.0f - 2.0f / (ex + 1.0f)));
                    }
                }
                if (p.gate) {
                    const float* gsel = (p.gate2 && ((m / p.rows_per_batch) & 1)) ? p.gate2 : p.gate;
#pragma unroll
                    for (int e = 0; e < VEC; e += 4) {
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gsel + nout0 + ncol + e);
#pragma unroll
                        for (int f = 0; f < 4; ++f) v[e + f] *= g4[f];
                    }
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += (float)rres[it][e];
                }
                V16 o16;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o16[e] = (T)v[e];
                *reinterpret_cast<V16*>(out + o) = o16;
            }
        }
    }
}

template <typename T, int BM, int BN, int MODE, bool GEGLU, int WM = 4, int WN = 1, int NST = 2>
int launch_one(const GemmArgs& a, hipStream_t s) {
    constexpr int NW = WM * WN;
    constexpr int LDS = gemm_lds_bytes<T, BM, BN, GEGLU, WM, WN, NST>();
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_done = false;   // one handle per device / one host thread per handle
    auto kern = gemm_kernel<T, BM, BN, MODE, GEGLU, WM, WN, NST>;
    if (!attr_done) {
        DSIM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done = true;
    }
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    // byte extents of the three operands for the buffer descriptors (32-bit offsets: < 2 GiB each)
    GemmArgs g = a;
    const size_t es = sizeof(T);
    size_t a0b, a1b = 16;
    if (MODE == GEMM_CONV3) {
        const size_t bimg = (size_t)a.M / ((size_t)a.Hout * a.Wout);
        a0b = bimg * a.Hin * a.Win * a.C0 * es;
    } else {
        a0b = (size_t)a.M * a.C0 * es;
        if (a.A1) a1b = (size_t)a.M * a.C1 * es;
    }
    const size_t wb = (size_t)a.N * a.K * es;
    if (a0b >= 0x7fffffffull || a1b >= 0x7fffffffull || wb >= 0x7fffffffull) return DSIM_ERR_INVALID;
    g.a0_bytes = (unsigned)a0b; g.a1_bytes = (unsigned)a1b; g.w_bytes = (unsigned)wb;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(NW * 64), LDS, s, g, tilesN);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

int check_args(const GemmArgs& a, int BK) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % BK) return DSIM_ERR_INVALID;
    if (a.mode == GEMM_CONV3) {
        if (a.C0 % BK || a.K != 9 * a.C0 || a.A1) return DSIM_ERR_INVALID;
    } else {
        if (a.C0 % BK || (a.A1 && a.C1 % BK) || a.K != a.C0 + (a.A1 ? a.C1 : 0)) return DSIM_ERR_INVALID;
    }
    if (a.epi == EPI_GEGLU && (a.mode != GEMM_LINEAR || a.N % 64)) return DSIM_ERR_INVALID;
    if (!a.zero_page) return DSIM_ERR_INVALID;
    return DSIM_OK;
}

template <typename T>
int launch_typed(const GemmArgs& a, hipStream_t s) {
    const int st = check_args(a, Traits<T>::BK);
    if (st != DSIM_OK) return st;
    int bm, bn;
    gemm_tile_choice(a, &bm, &bn);
    const bool big = bm == 256, n160 = bn == 160;
    (void)big;
    if constexpr (sizeof(T) == 2) {
        // bf16, big problems: 256-row tiles, 8 waves as 4(M) x 2(N), 64-row x (BN/2)-column sub-tiles
        if (big) {
            if (a.epi == EPI_GEGLU) return launch_one<T, 256, 256, GEMM_LINEAR, true, 4, 2>(a, s);
            if (a.mode == GEMM_CONV3)
                return bn == 320 ? launch_one<T, 256, 320, GEMM_CONV3, false, 4, 2>(a, s)
                                 : launch_one<T, 256, 256, GEMM_CONV3, false, 4, 2>(a, s);
            return bn == 320 ? launch_one<T, 256, 320, GEMM_LINEAR, false, 4, 2>(a, s)
                             : launch_one<T, 256, 256, GEMM_LINEAR, false, 4, 2>(a, s);
        }
    }
    if (a.epi == EPI_GEGLU) return launch_one<T, 128, 128, GEMM_LINEAR, true>(a, s);
    if (a.mode == GEMM_CONV3)
        return n160 ? launch_one<T, 128, 160, GEMM_CONV3, false>(a, s) : launch_one<T, 128, 128, GEMM_CONV3, false>(a, s);
    return n160 ? launch_one<T, 128, 160, GEMM_LINEAR, false>(a, s) : launch_one<T, 128, 128, GEMM_LINEAR, false>(a, s);
}

}  // namespace

int launch_gemm(const GemmArgs& a, int dtype, hipStream_t s) {
    if (dtype == DSIM_BF16) return launch_typed<bf16>(a, s);
    if (dtype == DSIM_F32) return launch_typed<float>(a, s);
    return DSIM_ERR_INVALID;
}

}  // namespace dsim
