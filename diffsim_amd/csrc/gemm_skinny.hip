// Small-batch implicit GEMM for gfx950: the same arithmetic as gemm_kernel (gemm.hip) for problems too small to fill the chip.
//
// The reference scores ONE pair per call (/root/reference/cute_main.py:111-132): 4 U-Net batch elements, so the 8x8 and 16x16
// levels are GEMMs of 256 / 1024 rows against K = 11520 .. 23040.  gemm_kernel's 128 x 160 tiles give 16-64 workgroups on 256
// CUs, and each walks its K tiles through a 2-stage LDS ring: one DMA round trip (~1 us) per 64-deep K tile, 195-390 us per
// conv whatever the batch (profiles/r04_small_batch.txt).  Splitting K would change the summation order, i.e. a pair's score
// would depend on the batch it was scored in.  This kernel keeps gemm_kernel's per-element arithmetic bit for bit -- the same
// v_mfma_f32_16x16x32 sequence over k, the same operand roles, the same rounding points -- and changes only what is invisible
// in the result:
//   * 64 x 64 tiles, 4 waves as 2 x 2 (32 x 32 per wave): 4-5x more workgroups, each streaming a 64-column weight slice;
//   * an 8-slot LDS ring (16 KB per K tile), seven K tiles in flight behind a counted vmcnt: a workgroup's K loop runs at its
//     CU's L2->LDS rate instead of one memory round trip per tile;
//   * no transpose epilogue (the problem is small): 8-byte stores straight from the D^T accumulators.
// launch_gemm (gemm.hip) routes here when gemm_kernel's grid would leave most of the chip idle.
#include "common.h"

namespace dsim {
namespace {

constexpr int SNW = 4, SBK_BYTES = 128;

// SBM x SBN tile, 4 waves as 2 x 2 (SBN a multiple of 32) or, for the 80-column tiles, as 4 x 1 (every wave SBM/4 rows x all columns);
// ring of SDEPTH slots (as many as fit in 144 KB, at most 8).  The weight rows are staged in 32-row groups (8 rows per wave): an
// 80-column tile stages 96, the last 16 as zero-filling out-of-range pieces so that every wave issues the same number of DMA
// pieces per K tile (the counted vmcnt needs one number).
template <int MODE, bool RES, int SBM, int SBN>
__global__ __launch_bounds__(SNW * 64, 1) void gemm_skinny_kernel(const GemmArgs p, const int tilesN) {
    constexpr int BK = 64, ES = 2;
    constexpr bool W41 = SBN % 32 != 0;                      // 4 x 1 wave layout
    constexpr int WLM = W41 ? 4 : 2, WLN = 4 / WLM;
    constexpr int SBNP = (SBN + 31) / 32 * 32;               // staged weight rows
    constexpr int SSTAGE = (SBM + SBNP) * SBK_BYTES;
    constexpr int SDEPTH = (144 * 1024 / SSTAGE) < 8 ? (144 * 1024 / SSTAGE) : 8;      // SDEPTH - 1 K tiles in flight
    constexpr int NAP = SBM / 32, NBP = SBNP / 32;           // 8-row DMA pieces per wave per K tile: activation, weight
    constexpr int SPW = NAP + NBP;
    constexpr int TM = SBM / (16 * WLM), TN = SBN / (16 * WLN);      // 16 x 16 accumulator tiles per wave
    static_assert(SBM % (16 * WLM) == 0 && SBN % (16 * WLN) == 0 && SBM % 32 == 0, "wave tile");
    static_assert(SDEPTH >= 3 && (SDEPTH - 2) * SPW <= 63, "ring");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the DMA pieces' LDS destinations need no v_readfirstlane)
    const int wm = W41 ? wave : wave >> 1, wn = W41 ? 0 : wave & 1;
    const int l15 = lane & 15, quad = lane >> 4, lrow = lane >> 3;
    const int wrow = wave * 8 + lrow;                                 // row inside a 32-row group of DMA pieces
    const unsigned celb = ((lane & 7) ^ ((wrow >> 1) & 7)) * 16;      // swizzled source chunk (gemm_kernel's LDS image)
    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A0, 0, (int)p.a0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A1 ? p.A1 : p.A0), 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);

    const int m0 = (blockIdx.x / tilesN) * SBM, n0 = (blockIdx.x % tilesN) * SBN;
    // the activation rows and the weight rows this lane stages per K tile
    int a_iy0[NAP], a_ix0[NAP];
    unsigned a_base[NAP], b_voff[NBP];
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
        const int m = m0 + i * 32 + wrow;
        if (MODE == GEMM_CONV3) {
            int b, oy, ox;
            if (p.lwo >= 0) {
                b = m >> p.lhw;
                const int rem = m & ((1 << p.lhw) - 1);
                oy = rem >> p.lwo; ox = rem & ((1 << p.lwo) - 1);
            } else {
                const int hw = p.Hout * p.Wout;
                b = m / hw;
                const int rem = m - b * hw;
                oy = rem / p.Wout; ox = rem - oy * p.Wout;
            }
            a_iy0[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
            a_ix0[i] = ox * p.stride - p.pad;
            a_base[i] = (unsigned)b * (unsigned)(p.Hin * p.Win);
        } else {
            a_iy0[i] = a_ix0[i] = 0;
            a_base[i] = (m < p.M) ? (unsigned)m : OOB;
        }
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int n = n0 + i * 32 + wrow;
        b_voff[i] = (n < p.N && i * 32 + wrow < SBN) ? (unsigned)n * (unsigned)p.K * (unsigned)ES + celb : OOB;
    }
    auto issue = [&](int t) {
        char* sa = smem + (t % SDEPTH) * SSTAGE;
        char* sb = sa + SBM * SBK_BYTES;
        const int k0 = t * BK;
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            unsigned vo;
            int soff;
            bool second = false;
            if (MODE == GEMM_CONV3) {
                const int tap = k0 / p.C0;
                soff = (k0 - tap * p.C0) * ES;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                const unsigned pix = a_base[i] + (unsigned)((iy >> p.ups) * p.Win + (ix >> p.ups));
                vo = ok ? pix * (unsigned)p.C0 * (unsigned)ES + celb : OOB;
            } else {
                second = k0 >= p.C0;
                soff = (second ? k0 - p.C0 : k0) * ES;
                vo = a_base[i] != OOB ? a_base[i] * (unsigned)(second ? p.C1 : p.C0) * (unsigned)ES + celb : OOB;
            }
            auto lds = (__attribute__((address_space(3))) void*)(sa + (i * SNW + wave) * 1024);
            if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, lds, 16, (int)vo, soff, 0, 0);
            else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, lds, 16, (int)vo, soff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NBP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(sb + (i * SNW + wave) * 1024), 16,
                                                     (int)b_voff[i], k0 * ES, 0, 0);
    };

    // D^T accumulators as in gemm_kernel: acc[i][j][r] = output row m0 + wm SBM/WLM + 16 i + l15, column n0 + wn SBN/WLN + 16 j + 4 quad + r.
    // Linear layers start at the bias, the 3x3 conv adds it in f32 behind the K loop (gemm_kernel's BIAS_INIT rule).
    f32x4 acc[TM][TN];
    constexpr bool BIAS_INIT = MODE == GEMM_LINEAR;
    const int nk = p.K / BK;
    bool odd[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) odd[i] = p.bias2 && (((m0 + wm * (SBM / WLM) + i * 16 + l15) / p.rows_per_batch) & 1);
    f32x4 b4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + wn * (SBN / WLN) + j * 16 + 4 * quad;
        b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && nb < p.N) {
            b4[j] = *reinterpret_cast<const f32x4*>(p.bias + nb);
            if (p.bias2) c4 = *reinterpret_cast<const f32x4*>(p.bias2 + nb);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = BIAS_INIT ? (odd[i] ? c4[e] : b4[j][e]) : 0.f;
    }
    // (the bias loads sit in front of the ring's DMA pieces: waiting for them does not drain the ring)
#pragma unroll
    for (int s = 0; s < SDEPTH - 1; ++s)
        if (s < nk) issue(s);
    const int foff0 = l15 * 128 + ((quad ^ ((l15 >> 1) & 7)) << 4);
    for (int t = 0; t < nk; ++t) {
        // K tile t has landed in this wave's pieces (the SDEPTH - 2 younger tiles may still fly; the last tiles drain the queue)
        if (t + SDEPTH - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SDEPTH - 2) * SPW) : "memory");
        else                     asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... in every wave's, and every wave is past its fragment reads of K tile t - 1 (each MFMA waited for its own): a bare
        // s_barrier -- __syncthreads() would put vmcnt(0) in front of it and drain the ring
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + SDEPTH - 1 < nk) issue(t + SDEPTH - 1);               // into the slot K tile t - 1 has left
        const char* sa = smem + (t % SDEPTH) * SSTAGE + wm * (SBM / WLM) * 128;
        const char* sb = smem + (t % SDEPTH) * SSTAGE + SBM * SBK_BYTES + wn * (SBN / WLN) * 128;
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            h16x8 xf[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const h16x8*>(sa + i * 16 * 128 + (foff0 ^ (step << 6)));
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const h16x8*>(sb + j * 16 * 128 + (foff0 ^ (step << 6)));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = H16_MFMA_16x16x32(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue: gemm_kernel's rounding points (bias in f32, round to the 16-bit type, then the residual add and a second round)
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.out), 0, (int)p.out_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (SBM / WLM) + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + wn * (SBN / WLN) + j * 16 + 4 * quad;
            const bool ok = m < p.M && nb < p.N;
            const unsigned off = ok ? ((unsigned)m * (unsigned)p.ldo + (unsigned)nb) * (unsigned)ES : OOB;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
            if (!BIAS_INIT) {
                f32x4 bb = b4[j];
                if (p.bias2 && odd[i] && nb < p.N) bb = *reinterpret_cast<const f32x4*>(p.bias2 + nb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bb[e];
            }
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (h16)v[e];
            if (RES) {
                const u32x2 rr = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rR, (int)off, 0, 0));
                const h16x4 r4 = __builtin_bit_cast(h16x4, rr);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (h16)((float)o[e] + (float)r4[e]);
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rO, (int)off, 0, 0);
        }
    }
}

template <int MODE, bool RES, int SBM, int SBN>
int launch_skinny_t(const GemmArgs& g, hipStream_t s) {
    constexpr int STG = (SBM + (SBN + 31) / 32 * 32) * SBK_BYTES;
    constexpr int LDS = ((144 * 1024 / STG) < 8 ? (144 * 1024 / STG) : 8) * STG;
    static DeviceOnce once;
    auto kern = gemm_skinny_kernel<MODE, RES, SBM, SBN>;
    CK_ONCE(once, kern, LDS);
    const int tilesM = (g.M + SBM - 1) / SBM, tilesN = (g.N + SBN - 1) / SBN;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(SNW * 64), LDS, s, g, tilesN);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

template <int MODE, bool RES>
int launch_skinny(const GemmArgs& g, hipStream_t s) {
    int bm, bn;
    gemm_skinny_tile(g, &bm, &bn);
    if (bm == 128 && bn == 64) return launch_skinny_t<MODE, RES, 128, 64>(g, s);
    if (bm == 64 && bn == 128) return launch_skinny_t<MODE, RES, 64, 128>(g, s);
    if (bm == 128 && bn == 128) return launch_skinny_t<MODE, RES, 128, 128>(g, s);
    if (bm == 64 && bn == 80) return launch_skinny_t<MODE, RES, 64, 80>(g, s);
    if (bm == 128 && bn == 80) return launch_skinny_t<MODE, RES, 128, 80>(g, s);
#ifdef DSIM_DEVTOOLS
    if (bm == 128 && bn == 160) return launch_skinny_t<MODE, RES, 128, 160>(g, s);     // kbench sweep only (never the heuristic's choice)
#endif
    return launch_skinny_t<MODE, RES, 64, 64>(g, s);
}

}  // namespace

#ifdef DSIM_DEVTOOLS
int g_skinny_tile = 0;          // kbench: 0 heuristic, else (bm << 8) | bn
#endif

// Tile of the small-batch kernel.  One workgroup runs per CU (the ring fills LDS) and streams (bm + bn) x 128 B per K tile through
// its CU's L2 -> LDS path, so the tile with the smallest bm + bn that still makes ONE round (<= CUs workgroups) wins -- by less than
// the byte count says once most CUs stream at the same time (the L2s' aggregate rate, ~8.7 TB/s, takes over).  Measured (tools/kbench
// KB_SKINNY=2, profiles/r04_small_batch.txt), ms at K = 11520: 1024 x 1280: 64x80 (256 workgroups) 0.094 | 64x128 (160) 0.098 | 128x64
// 0.105 | 128x128 (80) 0.133 | 64x64 (320: two rounds) 0.143 | gemm_kernel 0.124; 2048 x 1280: 128x80 (256) 0.123 | 128x128 (160) 0.135 |
// 64x128 (320) 0.192 | gemm_kernel 0.131; 256 / 512 x 1280: 64x64 0.072 against 0.124.
static long skinny_count(const GemmArgs& a, int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); }
static const int kSkinnyTiles[5][2] = {{64, 64}, {64, 80}, {64, 128}, {128, 80}, {128, 128}};      // by bm + bn
void gemm_skinny_tile(const GemmArgs& a, int* bm, int* bn) {
#ifdef DSIM_DEVTOOLS
    if (g_skinny_tile) { *bm = g_skinny_tile >> 8; *bn = g_skinny_tile & 255; return; }
#endif
    for (const auto& t : kSkinnyTiles)
        if (skinny_count(a, t[0], t[1]) <= cu_count() && (t[1] != 80 || a.N % 80 == 0)) { *bm = t[0]; *bn = t[1]; return; }
    *bm = 128; *bn = 128;
}

// Does the small-batch kernel take this problem?  Plain / residual epilogues of the 16-bit modes with a K loop long enough for the
// ring to matter, when gemm_kernel's grid (128-row tiles) would occupy at most a quarter of the CUs, or when its best tile makes
// one round of 0.6 ... 1 workgroups per CU (2048 x 1280 x 11520: 0.132 against 0.146 ms; 4096 x 640: equal).
bool gemm_skinny_applies(const GemmArgs& a) {
    if (a.epi == EPI_GEGLU || a.act != 0 || a.gate != nullptr || a.out_split) return false;
    if (a.K < 8 * 64 || a.K % 64 || a.C0 % 64 || (a.A1 && a.C1 % 64) || a.N % 8) return false;
    const long reg_tiles = (long)((a.M + 127) / 128) * ((a.N + 159) / 160);
#ifdef DSIM_DEVTOOLS
    if (g_gemm_skinny == 2) return reg_tiles <= 2 * cu_count();          // kbench: widen the rule for a sweep
#endif
    if (reg_tiles * 4 <= cu_count()) return true;
    int bm, bn;
    gemm_skinny_tile(a, &bm, &bn);
    const long c = skinny_count(a, bm, bn);
    return c * 10 >= (long)cu_count() * 6 && c <= cu_count();
}

// a: operand extents already filled in (launch_gemm does it)
int launch_gemm_skinny(const GemmArgs& g, hipStream_t s) {
    if (g.mode == GEMM_CONV3) return g.epi == EPI_RESIDUAL ? launch_skinny<GEMM_CONV3, true>(g, s) : launch_skinny<GEMM_CONV3, false>(g, s);
    return g.epi == EPI_RESIDUAL ? launch_skinny<GEMM_LINEAR, true>(g, s) : launch_skinny<GEMM_LINEAR, false>(g, s);
}

}  // namespace dsim
