// Implicit-GEMM MFMA kernel for gfx950: linear / 1x1 conv / 3x3 conv (stride 1|2, folded
// nearest-2x upsample, channel-concat of two sources) with fused bias / residual / GEGLU
// epilogues.  One template serves the h16 production path (v_mfma_f32_16x16x32_bf16) and the
// fp32 parity path (v_mfma_f32_16x16x4_f32, an exact f32 fma chain).  The 16x16 shapes, not the 32x32 ones: at equal cycles
// per FLOP the chip holds a ~12 % higher clock on them (MI355X_MICROARCH.md "DVFS give-back" item 7; measured here in the
// regime of this K loop by tools/ubench_mfma_shape.hip: 1.94 against 1.74 GHz, 1855 against 1660 TF/s).
//
// Replaces what the reference dispatches to cuDNN/cuBLAS through diffusers' ResnetBlock2D /
// Transformer2DModel / Attention / FeedForward modules (control flow
// in /root/reference/diffsim/hacked_modules.py:17-136, 261-434).
//
// Data layout: activations token-major [M][C]; weights packed [N][K] with K contiguous
// (conv: k = tap*Cin + c).  LDS tiles are [rows][128 B] with the 16-B chunk index XOR-swizzled
// by (row>>1)&7 so that ds_read_b128 fragment reads are bank-conflict free; tiles are filled by
// LDS-DMA (buffer_load_dwordx4 ... lds through buffer descriptors), whose lane-linear destination
// means the swizzle is applied to the per-lane SOURCE address.  Zero padding of conv borders and
// ragged M/N edges comes from an out-of-range buffer offset (the DMA then writes zeros).
//
// Schedule (details at gemm_kernel): persistent workgroups, 2-stage LDS ring per K tile, register
// double-buffered fragments with the K loop rotated by one piece, D^T accumulators transposed
// through wave-private LDS slabs into 16-byte coalesced stores, residual prefetched under the
// transpose, next tile's first stage in flight under the epilogue.
#include "common.h"

#include <algorithm>
#include <type_traits>

namespace dsim {

#ifdef DSIM_DEVTOOLS
int g_force_bm = 0;
int g_gemm_persistent = 1;
int g_gemm_exp = 0;
int g_gemm_skinny = 1;
unsigned long long* g_gemm_stamps = nullptr;
#endif
// Tile choice.  Small problems: 128-row tiles, 4 waves, two workgroups per CU (160-wide when N
// allows -- every SD channel count is a multiple of 160 -- else 128; 80-wide where 160 would leave one workgroup per CU).  h16
// problems with enough 256-row tiles to fill the chip: 256 x 320 (or 256 x 256, or 256 x 192 for the DiT widths) tiles, 8 waves as
// 4 x 2.  Every choice below was made by timing the neighbouring choice on the shapes it serves (profiles/r04_experiments.txt items
// 12-15, profiles/r04_small_batch.txt); gemm_launch_tile() maps the result onto the instantiations that exist.
void gemm_tile_choice(const GemmArgs& a, int* bm, int* bn) {
    const bool geglu = a.epi == EPI_GEGLU;
    if (geglu && a.geglu_blk == 16) {                      // 16-row [h | g] blocks: the 320 / 160-column tiles (N % 320 == 0)
        const long tiles = (long)((a.M + 255) / 256) * (a.N / 320);
        const bool big = g_force_bm ? g_force_bm == 256 : tiles >= 256;
        *bm = big ? 256 : 128;
        *bn = big ? 320 : 160;
        return;
    }
#ifdef DSIM_DEVTOOLS
    const bool n320 = !geglu && a.N % 320 == 0 && !(g_gemm_exp & 4096), n256 = a.N % 256 == 0;
#else
    const bool n320 = !geglu && a.N % 320 == 0, n256 = a.N % 256 == 0;
#endif
    // A ragged last 320-wide tile wasting <= 5 % of the columns (DiT: fused qkv N = 3456 = 10.8 tiles, Mlp.fc1 N = 4608 = 14.4) beats 256-wide tiles when the
    // persistent grid's rounds come out shorter: the 64 x 160 wave tile is 5-10 % faster per column (profiles/r04_experiments.txt
    // items 13, 15), a round is one tile per CU.  65536 x 3456 x 1152: 11 rounds of 320 against 14 of 256 (0.491 -> 0.467 ms);
    // at 32768 rows 5.5 -> 6 rounds against 7: the 256-wide tiles stay (0.223 against 0.234 ms).
    {
        const int c320 = (a.N + 319) / 320, c256 = (a.N + 255) / 256, tm = (a.M + 255) / 256, cus = cu_count();
#ifdef DSIM_DEVTOOLS
        const bool allow = !(g_gemm_exp & 16384) && !(g_gemm_exp & 4096);
#else
        const bool allow = true;
#endif
        const bool act_only = a.act == 1 && !a.gate && a.epi != EPI_RESIDUAL;      // the epilogue kinds instantiated at 256 x 320
        const long r320 = ((long)tm * c320 + cus - 1) / cus, r256 = ((long)tm * c256 + cus - 1) / cus;
        if (allow && !geglu && a.mode == GEMM_LINEAR && !n320 && (act_only || (!a.act && !a.gate)) && (long)tm * c320 >= cus &&
            (long)c320 * 320 * 20 <= (long)a.N * 21 && r320 * 320 * 19 <= r256 * 256 * 20 && g_force_bm != 128) {
            *bm = 256; *bn = 320;
            return;
        }
    }
    // 192-wide: the DiT widths (1152, 3456) that neither 320 nor 256 divides; linear layers only
    bool n192 = !geglu && a.mode == GEMM_LINEAR && !n320 && !n256 && a.N % 192 == 0;
    // ... unless a ragged last 256-wide tile wastes at most 5 % of the columns (DiT's fused qkv, N = 3456: 13.5 tiles): the
    // 64 x 128 wave tile reads 0.75 LDS fragments per MFMA against 0.83 for 64 x 96 (qkv projection 8.67 -> 7.83 ms per step)
    if (n192 && (long)((a.N + 255) / 256) * 256 * 20 <= (long)a.N * 21 && (long)((a.M + 255) / 256) * ((a.N + 255) / 256) >= 256) {
        *bm = g_force_bm == 128 ? 128 : 256;
        *bn = g_force_bm == 128 ? 128 : 256;
        return;
    }
    // 128-wide: the VAE's 128-channel 3x3 convs at 512 x 512 (N = 128 exactly)
    const bool n128 = !geglu && a.mode == GEMM_CONV3 && a.N == 128;
    int want256 = 0;
    if (n128) {
        const long tiles = (long)((a.M + 255) / 256);
        if (tiles >= 256 && g_force_bm != 128) { *bm = 256; *bn = 128; return; }
    }
    if (n320 || n256 || n192) {
        const int bnb = n320 ? 320 : (n256 ? 256 : 192);
        const long tiles = (long)((a.M + 255) / 256) * (a.N / bnb);
        want256 = tiles >= 256;
    }
    if (a.force_big) want256 = (n320 || n256 || n192);
    if (g_force_bm == 128) want256 = 0;                       // development override (kbench A/B)
    if (g_force_bm == 256) want256 = (n320 || n256 || n192);
    if (want256) { *bm = 256; *bn = n320 ? 320 : (n256 ? 256 : 192); return; }
    *bm = 128;
    *bn = (a.N % 160 == 0 && !geglu) ? 160 : 128;
    // 128 x 160 tiles that would leave ONE 4-wave workgroup per CU (<= CUs tiles: 4096 x 1280 = 16 batch elements at the 16 x 16
    // level): half-width tiles put two on every CU -- conv 4096 x 1280 x 11520 0.163 -> 0.143 ms, x 23040 0.312 -> 0.281; where 128 x 160
    // already gives two per CU they lose 35 % (profiles/r04_small_batch.txt)
#ifdef DSIM_DEVTOOLS
    const bool allow80 = !(g_gemm_exp & 2048);
#else
    const bool allow80 = true;
#endif
    if (allow80 && *bn == 160 && (long)((a.M + 127) / 128) * (a.N / 160) <= cu_count()) *bn = 80;
}

// The tile gemm_kernel is launched with (gemm_tile_choice + what the instantiation set allows): the f32 parity mode has
// 128-row tiles only (its GEGLU with 16-row blocks the 160-column one whatever the size); the gated DiT epilogues exist at
// 256 x 256, 256 x 192 and 128 x 128, the tanh-GELU-only one also at 256 x 320.  The executors name their profile families by it.
void gemm_launch_tile(const GemmArgs& a, int dtype, int* bm, int* bn) {
    gemm_tile_choice(a, bm, bn);
    const bool slow = a.act != 0 || a.gate != nullptr;
    const bool act_only = a.act == 1 && !a.gate && a.epi != EPI_RESIDUAL;
    // N = 128 convs, 16-bit: 512-row tiles (8 waves as 8 x 1, 64 x 128 per wave: 0.375 instead of 0.5 KB of fragment reads per MFMA; -6 % at
    // 512 x 512 x 128, profiles/r05_experiments.txt item 9) where ONE image alone makes >= 256 of them on a power-of-two map -- so the
    // choice, and with it every partial sum of the epilogue statistics, is the same at every batch size
    if (*bm == 256 && *bn == 128 && dtype != DSIM_F32 && a.mode == GEMM_CONV3 && !a.bias2 && g_force_bm != 256) {
        const long hw = (long)a.Hout * a.Wout;
        if (hw >= 512L * 256 && !(hw & (hw - 1)) && a.Wout > 0 && !(a.Wout & (a.Wout - 1)) && a.M % hw == 0) *bm = 512;
    }
    if (*bm == 512) return;
    bool big = *bm == 256 && dtype != DSIM_F32;
    if (big && slow && *bn == 320 && !act_only) big = false;
    if (!big) {
        // (a 320-column choice the f32 mode cannot run falls back to 160 columns, not 128: N % 320 == 0 there, and the one-launch
        //  tapped q | k | v needs the tile width to divide out_split, a multiple of 320)
        const bool n160 = *bn == 160 || (*bn == 320 && !slow) || (a.epi == EPI_GEGLU && a.geglu_blk == 16);
        const bool n80 = *bn == 80 && dtype != DSIM_F32 && !slow;          // (16-bit instantiations only; f32: the 160-column tile)
        *bm = 128;
        *bn = slow ? 128 : (n80 ? 80 : ((n160 || *bn == 80) ? 160 : 128));
    }
}

namespace {

// In-kernel phase stamps (cdna_hip_programming.md section 7, "In-kernel stamps"): a -DDSIM_STAMPS diagnostic build of tools/kbench
// only; the sums go to a buffer of their own (GemmArgs.stamps), never into an output.  No stamp executes in the product.
#if defined(DSIM_DEVTOOLS) && defined(DSIM_STAMPS)
#define STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DECL unsigned long long st_[7] = {0, 0, 0, 0, 0, 0, 0}, sa_[6] = {0, 0, 0, 0, 0, 0}
#define STAMP_ACC(i) sa_[i] += st_[(i) + 1] - st_[i]
#else
#define STAMP(t) do { } while (0)
#define STAMP_DECL do { } while (0)
#define STAMP_ACC(i) do { } while (0)
#endif

template <typename T> struct Traits;
template <> struct Traits<h16> {
    static constexpr int BK = 64;     // elements per 128-byte LDS row
    static constexpr int VEC = 8;     // elements per 16-byte chunk
};
template <> struct Traits<float> {
    static constexpr int BK = 32;
    static constexpr int VEC = 4;
};

template <typename T> struct Vec16T;
template <> struct Vec16T<h16> { typedef h16x8 type; };
template <> struct Vec16T<float> { typedef f32x4 type; };

// A fragment is 16 bytes per lane for both dtypes: lane (r = lane & 15, q = lane >> 4) holds elements k = 8q .. 8q+7 (h16) or
// k = 4q .. 4q+3 (f32) of row r of a 16-row block, i.e. 16-byte chunk q of the 64-byte K step.
__device__ __forceinline__ void load_frag(h16x8& f, const char* p) { f = *reinterpret_cast<const h16x8*>(p); }
__device__ __forceinline__ void load_frag(f32x4& f, const char* p) { f = *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void mma(const h16x8& a, const h16x8& b, f32x4& c) {
    c = H16_MFMA_16x16x32(a, b, c, 0, 0, 0);
}
// f32: the 16-deep step is 4 exact-f32 MFMAs of depth 4; MFMA e takes element e of every lane, so lane quarter q supplies
// k = 4q + e to both operands (any k pairing that is the same for A and B is a valid dot product).
__device__ __forceinline__ void mma(const f32x4& a, const f32x4& b, f32x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
}
template <typename T> struct FragOf { typedef h16x8 type; };
template <> struct FragOf<float> { typedef f32x4 type; };

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

// LDS: [stage buffer 0][spare][stage buffer 1].  The epilogue's wave-private transpose slabs live in the buffer the
// last K step read plus the spare, so the OTHER buffer can already receive the next tile's first K stage while the
// epilogue runs (persistent workgroups, see gemm_kernel).
template <typename T, int BM, int BN, bool GEGLU, int WM, int WN>
constexpr int gemm_epi_bytes() {
    return WM * WN * 16 * ((GEGLU ? BN / WN / 2 : BN / WN) * (int)sizeof(T) + 16);
}
template <typename T, int BM, int BN, bool GEGLU, int WM, int WN>
constexpr int gemm_spare_bytes() {
    const int stage = (BM + BN) * 128, epi = gemm_epi_bytes<T, BM, BN, GEGLU, WM, WN>();
    return epi > stage ? ((epi - stage + 1023) / 1024) * 1024 : 0;
}
// + the bias of the tile's columns.  16-bit kernels: one 1-KB slot per wave and tile parity, filled by ONE LDS-DMA piece per wave
// (the wave's own BN / WN columns) together with the tile's first K stage; f32: two slices of BN columns filled through registers.
template <typename T, int BM, int BN, bool GEGLU, int WM, int WN>
constexpr int gemm_lds_bytes() {
    // (512-row tiles: the two stages ARE the 160 KB; their bias comes straight from global memory into the accumulators)
    return 2 * (BM + BN) * 128 + gemm_spare_bytes<T, BM, BN, GEGLU, WM, WN>() +
           (BM >= 512 ? 0 : (sizeof(T) == 2 ? 2 * WM * WN * 1024 : 2 * BN * 4));
}

// WM x WN waves; each owns a (BM/WM) x (BN/WN) sub-tile of 16 x 16 accumulator tiles, so an activation fragment is reused by
// BN/WN/16 MFMAs and a weight fragment by BM/WM/16: with 64 x 160 per wave the LDS read traffic per 16-cycle MFMA is 0.35
// fragments (= 0.7 per 32 cycles) against 0.6 for 32 x 160 (the 4x1 layout) -- the LDS port, not the MFMA pipe, limits that one.
// Persistent workgroups: the grid is one (or two) workgroups per CU; each walks tiles vb = blockIdx.x + i * gridDim.x
// (XCD-aware order).  After a tile's K loop the next tile's first stage is issued into the free staging buffer
// BEFORE the epilogue, so its HBM latency and the epilogue's stores overlap instead of serialising per tile.
// EK (epilogue kind): EK_PLAIN bias only; EK_RES + residual; EK_SLOW the DiT epilogues (tanh-GELU activation, adaLN
// gate, optional residual decided at run time).  Compile-time, because a run-time residual flag makes hipcc keep
// every prefetched residual register in scratch, and the DiT math would add its register pressure to all users.
enum { EK_PLAIN = 0, EK_RES = 1, EK_SLOW = 2, EK_ACT = 3,     // EK_ACT: tanh-GELU only (DiT Mlp.fc1): no gate, no residual registers
       EK_PLAIN_GN = 4, EK_RES_GN = 5 };                        // + GroupNorm statistics of the output from the read-back (GemmArgs.gn_part)
template <typename T, int BM, int BN, int MODE, bool GEGLU, int WM, int WN, int EKT>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_kernel(const GemmArgs p, const int tilesN, const int ntiles, const int tilesM,
                                                               const int gn) {
    constexpr int NW = WM * WN;
    constexpr bool GNS = EKT >= EK_PLAIN_GN;               // epilogue statistics for the consumer's GroupNorm (the VAE's wide levels)
    constexpr int EK = GNS ? EKT - EK_PLAIN_GN : EKT;      // the epilogue kind proper
    constexpr bool CONV = MODE != GEMM_LINEAR;             // GEMM_CONV3 or GEMM_CONV3P (output map sides are powers of two)
    constexpr bool P2 = MODE == GEMM_CONV3P;
    constexpr int BK = Traits<T>::BK;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int TM = BM / (WM * 16), TN = BN / (WN * 16);        // 16 x 16 accumulator tiles per wave
    // weight fragments per piece of the K loop (a piece = PS x TM MFMAs between two pinned read groups)
    constexpr int PS = TM >= 4 ? 2 : (TN % 5 == 0 ? 5 : 4);
    constexpr int NP = TN / PS;                                    // pieces per 64-byte K step
    static_assert(TN % PS == 0 && (GEGLU ? TN % 2 == 0 : true), "wave tile");
    // GEGLU weights alternate [h | g] blocks of 32 packed rows (accumulator tiles [h, h, g, g]: tile j pairs with j + 2) or, for the
    // wave widths that are not a multiple of 64 columns (160, 80), of 16 rows (tiles [h, g]: j pairs with j + 1)
    constexpr bool G16 = GEGLU && TN % 4 != 0;
    constexpr int WBN = BN / WN;                           // columns per wave
    constexpr int NA = BM / (NW * 8);                      // 8-row DMA pieces of A per wave per stage
    constexpr int NBP = BN / 8;                            // 8-row DMA pieces of B per stage (all waves)
    constexpr int NB = (NBP + NW - 1) / NW;                // ... per wave (the last one may be partial)
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int SPARE = gemm_spare_bytes<T, BM, BN, GEGLU, WM, WN>();
    typedef typename FragOf<T>::type Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
#ifdef DSIM_DEVTOOLS
    // kbench ablations (timing only, outputs wrong): KB_GEXP bit 16 = residual loads dropped, bit 32 = output stores dropped
    // (zero-record descriptors: the range check drops the access, the instruction stream stays)
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (p.exp & 32) ? 0 : (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.out), 0, (p.exp & 16) ? 0 : (int)p.out_bytes, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.out), 0, (int)p.out_bytes, 0x00020000);
#endif

    // ---- per-thread state of the tile being staged ---------------------------------------------
    int m0 = 0, n0 = 0;
    int a_iy0[NA], a_ix0[NA];
    unsigned a_base[NA], a_voff[NA];
    unsigned b_voff[NB];
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    auto setup = [&](int vb) {
        // XCD-aware bijective remap: the 8 XCDs take consecutive block ids round-robin (gridDim.x is a multiple
        // of 8 whenever a workgroup walks more than one tile); give each XCD a contiguous run of logical tiles, and order
        // the logical tiles in column BANDS of gn tile columns, row panel by row panel inside a band: the ~32 workgroups an
        // XCD runs at one time then cover (32 / gn) activation panels x gn weight tiles instead of one panel x 32 weight
        // tiles, and the band's weight tiles stay in the XCD's 4 MiB L2 for the whole sweep down M (gemm_band_width()).
        const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7, slot = vb >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        const int band = bid / (gn * tilesM), rem = bid - band * (gn * tilesM);
        const int bw = min(gn, tilesN - band * gn);            // the last band may be narrower
        const int mt = rem / bw;
        m0 = mt * BM;
        n0 = (band * gn + (rem - mt * bw)) * BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + i * (NW * 8) + wrow;
            if (CONV) {
                // output pixel -> (image, row, column): shifts when the output map's sides are powers of two (every SD level at
                // the sizes the 256-row tiles serve), else two integer divisions per row -- setup() runs twice per tile, outside
                // the MFMA shadow, and the divisions were ~440 of its VALU instructions per wave
                int b, oy, ox;
                if constexpr (P2) {
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
                a_voff[i] = OOB;
            } else {
                a_iy0[i] = a_ix0[i] = 0;
                a_base[i] = (m < p.M) ? (unsigned)m : OOB;
                a_voff[i] = (m < p.M) ? (unsigned)m * (unsigned)p.C0 * (unsigned)sizeof(T) + celb : OOB;
            }
        }
        // batched weights (the VAE's per-image q k^T and P v): the tile's rows belong to ONE batch (wb_rows % BM == 0)
        const unsigned wofs = (!CONV && p.wb_rows) ? (unsigned)(m0 / p.wb_rows) * p.wb_stride : 0u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + i * (NW * 8) + wrow;
            b_voff[i] = (n < p.N) ? wofs + (unsigned)n * (unsigned)p.K * (unsigned)sizeof(T) + celb : OOB;
        }
    };
    // B pieces this wave really issues per stage (wave-uniform): the tail piece exists only for
    // the first NBP % NW waves
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool b_tail = (NBP % NW == 0) || wave_u < (NBP % NW);
    const int wm_u = wave_u / WN, wn_u = wave_u - wm_u * WN;       // the wave's position in the WM x WN layout, as scalars
    // 16-bit kernels: the f32 bias of the wave's own BN / WN columns arrives by one LDS-DMA piece per wave and tile (issued with the
    // tile's first K stage, i.e. a whole epilogue ahead; columns beyond N read as zeros through the descriptor's range check, as
    // does everything when there is no bias).  The per-CFG-half bias2 of SDXL's resnets keeps the register path.
    constexpr bool BDMA = sizeof(T) == 2 && BM < 512;
    static_assert(!BDMA || (BN / WN) * 4 <= 1024, "bias slot");
    const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : p.W), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    auto bias_dma = [&](int par) {       // slice of the tile setup() last ran for, into the slot of tile parity `par`
        if (BDMA && !p.bias2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rBias, (__attribute__((address_space(3))) void*)(smem + 2 * STAGE + SPARE + (par * NW + wave_u) * 1024), 16,
                                                     (int)((unsigned)(n0 + wn_u * (BN / WN)) * 4u + (unsigned)lane * 16u), 0, 0, 0);
    };

    // K-tile t of the tile being staged: derive() refreshes the per-row offsets at a conv tap / second-source
    // boundary, issue() launches the LDS-DMA pieces of this wave into staging buffer `buf`
    auto derive = [&](int t) {
        const int k0 = t * BK;
        if (CONV) {
            const int tap = k0 / p.C0;
            if (k0 == tap * p.C0) {   // first K tile of a tap: re-derive the gathered pixel of every row
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                    const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                    const unsigned pix = a_base[i] + (unsigned)((iy >> p.ups) * p.Win + (ix >> p.ups));
                    a_voff[i] = ok ? pix * (unsigned)p.C0 * (unsigned)sizeof(T) + celb : OOB;
                }
            }
        } else if (k0 == p.C0) {      // first K tile of the concatenated second source
#pragma unroll
            for (int i = 0; i < NA; ++i)
                a_voff[i] = a_base[i] != OOB ? a_base[i] * (unsigned)p.C1 * (unsigned)sizeof(T) + celb : OOB;
        }
    };
    // (DMAW: the LDS destination of a DMA piece is wave-uniform -- formed from the scalar wave index, M0 is then one s_add per piece
    //  instead of v_add + v_readfirstlane + s_mov: 1-2 % on the convs, profiles/r04_experiments.txt item 7)
#define DMAW wave_u
    auto issue = [&](int t, int buf) {
        char* sa = smem + (buf ? STAGE + SPARE : 0);
        char* sb = sa + A_BYTES;
        const int k0 = t * BK;
        int soff;
        bool second = false;
        if (CONV) {
            const int tap = k0 / p.C0;
            soff = (k0 - tap * p.C0) * (int)sizeof(T);
        } else {
            second = k0 >= p.C0;
            soff = (second ? k0 - p.C0 : k0) * (int)sizeof(T);
        }
#ifdef DSIM_DEVTOOLS
        // kbench ablation (timing only): KB_GEXP bit 64 = the activation pieces of two tap columns out of three are not issued -- the
        // DMA count of a conv whose three horizontal taps share one staged row block; bit 128 = none of them after the first K tile
        const bool skipA = CONV && (((p.exp & 64) && (k0 / p.C0) % 3 != 0) || ((p.exp & 128) && t > 0));
#else
        constexpr bool skipA = false;
#endif
        if (!skipA) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            auto lds = (__attribute__((address_space(3))) void*)(sa + (i * NW + DMAW) * 1024);
            if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, lds, 16, (int)a_voff[i], soff, 0, 0);
            else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, lds, 16, (int)a_voff[i], soff, 0, 0);
        }
        }
        const int soffw = k0 * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (i + 1 < NB || b_tail)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(sb + (i * NW + DMAW) * 1024),
                                                         16, (int)b_voff[i], soffw, 0, 0);
    };
    auto stage = [&](int t, int buf) { derive(t); issue(t, buf); };

    f32x4 acc[TM][TN];
    const int nk = p.K / BK;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int quad = lane >> 4;                      // lane quarter: 16-byte chunk of the 64-byte K step
    // swizzled chunk offsets of the two K steps of a tile: chunk c = 4 s + quad, stored at c ^ ((row >> 1) & 7) with
    // row = 16 x + (lane & 15), i.e. (lane >> 1) & 7; step 1 is step 0 with bit 2 of the chunk index flipped
    const int foff0 = (lane & 15) * 128 + ((quad ^ ((lane >> 1) & 7)) << 4);

    // D^T tiles: acc[i][j] = W rows (n = 16 j + 4 quad + r, registers) x activation rows (m = 16 i + (lane & 15), lane).
    // Per 64-byte K step the TM activation fragments xf are held and the TN weight fragments stream through a two-piece
    // register ring wf (PS fragments per piece), the reads of piece q+1 pinned in front of the MFMAs of piece q.  The loop is
    // ROTATED by one piece: the MFMAs of a K tile's LAST piece are issued after the barrier that ends the tile, right behind
    // the first fragment reads of the NEXT tile -- so the matrix pipe has register-resident work while those reads (and the
    // barrier skew) are outstanding, instead of both waves of a SIMD idling on LDS latency at every tile start.
    Frag xf[2][TM], wf[2][PS];
    auto load_x = [&](int buf, int step, int set) {
        const char* sa = smem + (buf ? STAGE + SPARE : 0) + wm * (BM / WM) * 128 + (foff0 ^ (step << 6));
#pragma unroll
        for (int i = 0; i < TM; ++i) load_frag(xf[set][i], sa + i * 16 * 128);
    };
    auto load_w = [&](int buf, int q, int set) {         // piece q of the tile: step q / NP, weight tiles PS (q % NP) ...
        const int step = q / NP, j0 = (q - step * NP) * PS;
        const char* sb = smem + (buf ? STAGE + SPARE : 0) + A_BYTES + (wn * WBN + j0 * 16) * 128 + (foff0 ^ (step << 6));
#pragma unroll
        for (int j = 0; j < PS; ++j) load_frag(wf[set][j], sb + j * 16 * 128);
    };
    auto mma_piece = [&](int q) {
        const int step = q / NP, j0 = (q - step * NP) * PS;
#pragma unroll
        for (int j = 0; j < PS; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) mma(wf[q & 1][j], xf[step][i], acc[i][j0 + j]);
    };

    constexpr int ES = sizeof(T);
    constexpr int OUTW = GEGLU ? WBN / 2 : WBN;        // output columns this wave produces
    constexpr int RSO = OUTW * ES + 16;                 // staging row stride (bytes)
    constexpr int CPR = OUTW * ES / 16;                 // 16-byte chunks per output row
    const int Nout = GEGLU ? (p.N >> 1) : (p.out_split ? p.out_split : p.N);
    typedef typename Vec16T<T>::type V16;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NIT = (16 * CPR + 63) / 64;          // read-back iterations per 16-row slab
    constexpr bool SLOW = EK == EK_SLOW;
    constexpr bool ACT = EK == EK_ACT;
    // linear layers start their accumulators at the bias (see the tile loop); the 3x3 conv keeps the bias add in its epilogue
    // (its K loop is long enough that the epilogue's loads do not matter, and the changed register allocation cost it 2.5 %)
    constexpr bool BIAS_INIT = !CONV || BM >= 512;
    const bool has_res = EK == EK_RES || (SLOW && p.epi == EPI_RESIDUAL);

    STAMP_DECL;
#ifdef DSIM_DEVTOOLS
    // kbench experiment (KB_GEXP bits 8..11 = d): workgroups of odd XCD-slot parity start d x 4 us late, so that their epilogues
    // (the tile's HBM phase) fall into the other half's K loops
    if ((p.exp >> 8) & 15) {
        if ((blockIdx.x >> 3) & 1) {
            const int n = ((p.exp >> 8) & 15) * 2;         // s_sleep 127 ~ 8128 cycles ~ 2 us at 2 GHz... x 2 per unit
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
#endif
    int vb = blockIdx.x;
    int tile_par = 0;                                   // parity of this workgroup's tile counter (bias buffer)
    int b0 = 0;                                         // staging buffer holding K stage 0 of the current tile
    setup(vb);
    bias_dma(0);
    stage(0, 0);
    bool first = true;
    // INVARIANT of the counted wait at the top of a tile (s_waitcnt vmcnt(NST) below): behind the next tile's bias_dma + stage(0)
    // EVERY wave issues EXACTLY NST = TM x NIT vector-memory operations per epilogue -- the output stores of its TM slabs x NIT
    // pieces, unconditionally (edge pieces carry an out-of-range offset, they are not skipped); the residual LOADS of an epilogue are
    // all waited for (their data is used) before its last store is issued; loads and stores retire in order (vmcnt on gfx9).  An
    // epilogue form that makes a store conditional, or adds a vector-memory operation behind the stores, must change NST with it:
    // the K loop would otherwise read LDS before the DMA landed.  (-DDSIM_DEVTOOLS builds: KB_GEXP bit 32768 waits vmcnt(0) instead,
    // for A/B and for checking a new epilogue form against the uncounted wait.)
    constexpr int NST = TM * NIT;                       // stores a wave issues per epilogue
    static_assert(NST <= 60, "counted vmcnt");
    while (true) {
        STAMP(st_[0]);
        // 3x3 conv: the tile's bias slice goes to LDS here (80 lanes, one float4 each); the epilogue's register phase then reads
        // it with ds_read_b128 instead of 40 dependent global loads per tile, each behind a vmcnt(0) that also waited for the
        // next tile's first stage.  Buffer = tile parity: a wave is at most one tile ahead of the slowest (the barrier below).
        // (bias2, the per-CFG-half bias of SDXL's resnets, keeps the global path.)
        // bias of this tile in LDS, indexed by the column relative to the tile's first (16-bit: the wave's own slot, shifted so that
        // the same index works)
        float* const bias_lds = BDMA ? reinterpret_cast<float*>(smem + 2 * STAGE + SPARE + (tile_par * NW + wave_u) * 1024) - wn_u * WBN
                                     : reinterpret_cast<float*>(smem + 2 * STAGE + SPARE) + (tile_par ? BN : 0);
        // (the 16-bit residual epilogue reads its bias from LDS unconditionally: zeros when there is none; check_args refuses bias2 there)
        const bool bias_in_lds = !BIAS_INIT && (BDMA ? !p.bias2 : (p.bias && !p.bias2));
        if (!BDMA && bias_in_lds && tid < BN / 4) {
            const int n = n0 + tid * 4;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
            *reinterpret_cast<f32x4*>(bias_lds + tid * 4) = b4;
        }
        // 16-bit: this wave's pieces of stage 0 and its bias slice have landed once all but the previous epilogue's NST stores are
        // done (they were issued before them): the stores' acknowledgements drain under K tile 0 instead of being waited for here
        if constexpr (BDMA) {
#ifdef DSIM_DEVTOOLS
            if (first || (p.exp & 32768)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
            first = false;
        }
        if (BIAS_INIT && p.bias) {
            int ilane = lane;
            asm volatile("" : "+v"(ilane));                     // per-tile lane id: nothing derived from it stays live in the K loop
            const int iquad = ilane >> 4, il15 = ilane & 15;
            bool odd[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                odd[i] = p.bias2 && (((m0 + wm * (BM / WM) + i * 16 + il15) / p.rows_per_batch) & 1);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + wn * WBN + j * 16 + 4 * iquad;
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, c4 = {0.f, 0.f, 0.f, 0.f};
                if (BDMA && !p.bias2) {
                    b4 = *reinterpret_cast<const f32x4*>(bias_lds + (nb - n0));
                } else if (nb < p.N) {
                    b4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
                    if (p.bias2) c4 = *reinterpret_cast<const f32x4*>(p.bias2 + nb);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = odd[i] ? c4[e] : b4[e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
        }
        // stage 0 has landed in every wave; every wave is past the previous tile's epilogue (its LDS slabs are free)
        STAMP(st_[1]);
        if constexpr (BDMA) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        STAMP(st_[2]);
        load_x(b0, 0, 0);
        load_w(b0, 0, 0);
        for (int t = 0; t < nk; ++t) {
            const int cur = b0 ^ (t & 1);
            if (t + 1 < nk) stage(t + 1, cur ^ 1);
#pragma unroll
            for (int q = 0; q + 1 < 2 * NP; ++q) {
                if (q == 0) load_x(cur, 1, 1);                  // the second K step's activation fragments, a whole step ahead
                load_w(cur, q + 1, (q + 1) & 1);
                // pin the order "reads of piece q+1, then the MFMAs of piece q": left alone, hipcc sinks each ds_read to
                // just before its MFMA and every MFMA then waits out LDS latency
                __builtin_amdgcn_sched_barrier(0);
                mma_piece(q);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef DSIM_DEVTOOLS
            // kbench ablations (timing only): KB_GEXP bit 256 = no barrier at the end of a K tile, 512 = no wait for the staged pieces either
            if (!(p.exp & 512)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(p.exp & 256)) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                    // (also drains this wave's fragment reads of `cur`)
#endif
            if (t + 1 < nk) { load_x(cur ^ 1, 0, 0); load_w(cur ^ 1, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
            mma_piece(2 * NP - 1);                              // the tile's last piece, from registers
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(st_[3]);
        const int xbuf = b0 ^ ((nk - 1) & 1);          // buffer the last K step read: epilogue scratch = it + spare
        const int em0 = m0, en0 = n0;
        const int nvb = vb + (int)gridDim.x;
        const bool more = nvb < ntiles;
        if (more) {                                     // next tile's first stage flies during the epilogue
            setup(nvb);
            bias_dma(tile_par ^ 1);
            stage(0, xbuf ^ 1);
        }
        STAMP(st_[4]);
    // ---- epilogue -------------------------------------------------------------------------
    // The accumulators hold D^T: lane = output row m (lane & 15), the four registers = output columns
    // n = 4 (lane >> 4) + r of a 16-wide block.  Each wave transposes one 16-row slab at a time
    // through a private LDS region (the K loop's last barrier freed the buffer it lives in), then
    // streams it out row-contiguously: 16-byte coalesced residual loads and stores.
        // an opaque per-tile copy of the lane id: everything the epilogue derives from it would otherwise be
        // hoisted out of the persistent tile loop and held in registers through the K loop (spills)
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int el15 = elane & 15, equad = elane >> 4;
        char* const wst = smem + (xbuf ? STAGE : 0) + wave * (16 * RSO);
        const int nw0 = en0 + wn * WBN;                    // first packed weight row of this wave
        // out_split: the output columns are cut into equal runs that go to separate tensors (the tapped q | k | v projection as one
        // launch): tensor en0 / split, column en0 % split; BN divides the split, so a tile never straddles two tensors.  The
        // store descriptor is rebuilt per tile with the tensor's base (plain projection epilogue only: no residual, no GEGLU).
        int osel = 0;
        if (!GEGLU && EK == EK_PLAIN && MODE == GEMM_LINEAR && p.out_split) osel = __builtin_amdgcn_readfirstlane(en0 / p.out_split);
        const __amdgpu_buffer_rsrc_t rOt = (!GEGLU && EK == EK_PLAIN && MODE == GEMM_LINEAR)
            ? __builtin_amdgcn_make_buffer_rsrc((char*)p.out + (size_t)osel * p.out_split_stride, 0, (int)p.out_bytes, 0x00020000) : rO;
        const int nout0 = (GEGLU ? (nw0 >> 1) : nw0) - osel * p.out_split;
        const int mw0 = em0 + wm * (BM / WM);
        // Addresses of the read-back pieces (round 5).  Piece `it` of slab i sits at
        //     [uniform: ((mw0 + 16 i) ldo + nout0) ES]  +  [per lane, the same for every slab of the tile: (row(it) ldo + 8 c(it)) ES]
        // so the uniform part rides in the buffer instruction's SCALAR offset and the lane part is formed once per tile: no vector
        // arithmetic per piece (it was ~12 VALU per residual load and again per store: 240-480 per wave per tile, in an epilogue no
        // MFMA overlaps).  Edges: on gfx950 the scalar offset takes part in the buffer range check (tools/soff_probe.hip), and the
        // output / residual descriptors end at row M, so rows beyond M are dropped by the hardware; columns beyond N and the
        // pieces beyond the slab's 16 rows carry an out-of-range lane part.
        const int mw0_u = em0 + wm_u * (BM / WM);
        const int nout0_u = (GEGLU ? ((en0 + wn_u * WBN) >> 1) : (en0 + wn_u * WBN)) - osel * p.out_split;
        // (unsigned: for an output just under the 2 GiB gemm_fill_extents accepts, the byte offset of the rows past M of a ragged last
        //  tile exceeds INT_MAX -- it must wrap to an out-of-range offset, not be undefined)
        const unsigned sbase = (unsigned)(mw0_u * p.ldo + nout0_u) * (unsigned)ES;      // scalar offset of slab 0
        const unsigned sslab = 16u * (unsigned)p.ldo * (unsigned)ES;                    // ... + i * sslab
        unsigned lp[NIT];                                               // lane part of piece it
        int lrd[NIT];                                                   // its 16 bytes in the wave's LDS slab
        {
            int plane = lane;
            asm volatile("" : "+v"(plane));
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = plane + it * 64;
                const int row = idx / CPR, c = idx - row * CPR;
                lp[it] = (idx < 16 * CPR && nout0 + c * VEC < Nout) ? (unsigned)(row * p.ldo + c * VEC) * (unsigned)ES : OOB;
                lrd[it] = (row < 16 ? row : 15) * RSO + c * 16;         // rows beyond the slab read a neighbouring row; their store is dropped
            }
        }
        auto piece_v = [&](int, int, int it) -> unsigned { return lp[it]; };
        auto piece_s = [&](int i) -> int { return (int)(sbase + (unsigned)i * sslab); };
        // STORES add the scalar part to the vector offset instead (one v_add per piece): a buffer_store_dwordx4 with an SGPR soffset
        // reads its data registers late on gfx950, and hipcc (which pads the next VALU write of a wide store's data registers only
        // for a constant soffset) then lets e.g. the next piece's address arithmetic overwrite the first data dword -- seen as
        // address-like garbage in the first 4 bytes of 16-byte pieces, on the later-dispatched waves, timing dependent.
        auto store_v = [&](int i, int it) -> unsigned { return lp[it] == OOB ? OOB : lp[it] + sbase + (unsigned)i * sslab; };
        // residual prefetch: slab i's 16-byte pieces are requested before slab i is transposed, so the HBM latency
        // hides under the register phase instead of serialising the stores (at most 10 pieces in flight per lane:
        // the f32 parity mode would spill with all 20)
        constexpr int PF0 = TM > 1 ? (NIT + 1) / 2 : (NIT < 10 ? NIT : 10);
        u32x4 rres[TM][NIT];
        auto prefetch = [&](int i, int it0, int it1) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (it >= it0 && it < it1)
                    rres[i][it] = __builtin_amdgcn_raw_buffer_load_b128(rR, (int)piece_v(elane, i, it), piece_s(i), 0);
        };
        // 16-bit residual epilogue (round 5): the WHOLE tile's residual is requested up front.  The accumulators are rounded to the
        // 16-bit type first (their values leave through LDS in that type anyway: same rounding points as the slab-by-slab form), which
        // halves their registers; the freed half takes all TM x NIT residual pieces in flight at once, so the tile pays the HBM
        // latency of its residual once, under the conversions, instead of once per 16-row slab (profiles/r05_experiments.txt item 2:
        // the slab-by-slab form was 31-42 % of a K = N <= 1280 launch against 12-17 % for the plain epilogue).
        constexpr bool RES16 = EK == EK_RES && sizeof(T) == 2 && !GEGLU && CONV;      // (linears: the slab-by-slab form measured 1-3 % faster)
        // GroupNorm statistics of the STORED values (round 5, the VAE's 512 x 512 and 256 x 256 levels): a lane of the read-back
        // always holds the same 16-byte chunk column (64 % CPR == 0), so it sums its two 4-channel quads over all its rows of the
        // tile -- (sum, sum of squares) x 2 -- the lanes that share a column are folded by a fixed xor tree, and the wave's CPR
        // first lanes write 2 x (sum, sumsq) each: one partial per (wave's 64 rows, quad), in a fixed place.  A fixed order per tile
        // shape, and the tile shape is fixed per image size (the caller only asks for it where one image alone fills the chip), so the
        // statistics are the same bits at every batch size.
        float gq[4] = {0.f, 0.f, 0.f, 0.f};                // quad 0 sum, quad 1 sum, quad 0 sumsq, quad 1 sumsq
        auto gn_acc = [&](const V16& o) {
            if constexpr (GNS) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float f = (float)o[e];
                    gq[e >> 2] += f;
                    gq[2 + (e >> 2)] = fmaf(f, f, gq[2 + (e >> 2)]);
                }
            }
        };
        if constexpr (RES16) {
            h16x4 pk[TM][TN];
            constexpr int HP = TM >= 2 ? TM / 2 : 1;               // slabs per half: pack a half, request its residual, pack the other half
#pragma unroll
            for (int h0 = 0; h0 < TM; h0 += HP) {
                // (an opaque offset per half: hipcc otherwise reads the ten bias vectors once and spills them across the residual loads)
                int boff = (wn * WBN + 4 * equad) * 4;
                asm volatile("" : "+v"(boff));
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (!BIAS_INIT) b4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(bias_lds) + boff + j * 64);
#pragma unroll
                    for (int i = h0; i < h0 + HP; ++i) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = BIAS_INIT ? acc[i][j][e] : acc[i][j][e] + b4[e];
                        const h16x2 lo = __builtin_convertvector((f32x2){v[0], v[1]}, h16x2), hi = __builtin_convertvector((f32x2){v[2], v[3]}, h16x2);
                        pk[i][j] = (h16x4){lo[0], lo[1], hi[0], hi[1]};
                        // pinned here: left alone hipcc sinks the conversions below the residual loads and keeps the f32 accumulators
                        // (and, spilled, the bias vectors) live across them
                        asm volatile("" : "+v"(pk[i][j]) :: "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = h0; i < h0 + HP; ++i) prefetch(i, 0, NIT);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<h16x4*>(wst + el15 * RSO + (j * 16 + 4 * equad) * ES) = pk[i][j];
                int slane = lane;
                asm volatile("" : "+v"(slane));
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const V16 t = *reinterpret_cast<const V16*>(wst + lrd[it]);
                    const V16 r = __builtin_bit_cast(V16, rres[i][it]);
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    V16 o16;
#pragma unroll
                    for (int e = 0; e < VEC; e += 2) {
                        const h16x2 pr = __builtin_convertvector((f32x2){(float)t[e] + (float)r[e], (float)t[e + 1] + (float)r[e + 1]}, h16x2);
                        o16[e] = pr[0]; o16[e + 1] = pr[1];
                    }
                    if (lp[it] != OOB) gn_acc(o16);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o16), rO, (int)store_v(i, it), 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            __builtin_amdgcn_sched_barrier(0);      // keep slab i+1's loads from being hoisted above slab i (spills)
            if (has_res) prefetch(i, 0, PF0);
            __builtin_amdgcn_sched_barrier(0);
            const int slab_half_div = (SLOW && p.gate2) ? (mw0 + i * 16) / p.rows_per_batch : 0;
            const bool odd_half = !BIAS_INIT && p.bias2 && (((mw0 + i * 16 + el15) / p.rows_per_batch) & 1);
            // register phase: the D^T -> row-major transpose through LDS (linear mode: the bias is already in the accumulators)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // GEGLU: packed weight rows alternate 32-row blocks [h-block, g-block] = tiles [h, h, g, g]: tile j (j & 2 == 0)
                // pairs with tile j + 2, same lane, same register (16-row blocks: tiles [h, g], j pairs with j + 1)
                if (GEGLU && (j & (G16 ? 1 : 2))) continue;
                float v[4];
                if (GEGLU) {
                    constexpr int JG = G16 ? 1 : 2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * (sizeof(T) == 2 ? gelu_fast(acc[i][j + JG][e]) : gelu_erf(acc[i][j + JG][e]));
                } else if (BIAS_INIT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
                } else {
                    const int nb = nw0 + j * 16 + 4 * equad;
                    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                    if (bias_in_lds) {
                        b4 = *reinterpret_cast<const f32x4*>(bias_lds + (nb - en0));
                    } else if (p.bias && nb < p.N) {
                        b4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
                        if (p.bias2) {
                            const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.bias2 + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) b4[e] = odd_half ? c4[e] : b4[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[e];
                }
                const int col = (GEGLU ? (G16 ? (j >> 1) : ((j >> 2) * 2 + (j & 1))) : j) * 16 + 4 * equad;
                char* dst = wst + el15 * RSO + col * ES;
                if constexpr (sizeof(T) == 2) {
                    // two packed conversions (element-wise casts compiled to three v_cvt_pk + v_perm + v_alignbit per four values)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const h16x2 lo = __builtin_convertvector((f32x2){v[0], v[1]}, h16x2), hi = __builtin_convertvector((f32x2){v[2], v[3]}, h16x2);
                    const h16x4 pk = {lo[0], lo[1], hi[0], hi[1]};
                    *reinterpret_cast<h16x4*>(dst) = pk;
                } else {
                    f32x4 pk = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(dst) = pk;
                }
            }
            if (has_res && PF0 < NIT) prefetch(i, PF0, NIT);
            // read-back phase (same wave: LDS operations of one wave execute in order).  The store offsets are
            // re-derived from a fresh opaque lane id rather than kept in registers since the prefetch.
            int slane = lane;
            asm volatile("" : "+v"(slane));
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const V16 t = *reinterpret_cast<const V16*>(wst + lrd[it]);
                if (!SLOW && !ACT && !has_res) {         // plain projection: LDS -> HBM copy
                    if (lp[it] != OOB) gn_acc(t);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rOt, (int)store_v(i, it), 0, 0);
                    continue;
                }
                float v[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = (float)t[e];
                if (ACT || (SLOW && p.act == 1)) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        // tanh-GELU: 0.5 x (1 + tanh(u)) = x / (1 + e^(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3); with the
                        // constants folded -2u log2(e) = x (a + b x^2): 5 plain VALU + v_exp + v_rcp per element (an IEEE
                        // divide alone expands to ~10 instructions, and this epilogue is issue-bound)
                        const float x = v[e];
                        const float t = x * fmaf(-0.10294324f, x * x, -2.3022082f);
                        v[e] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
                    }
                }
                if (SLOW && p.gate) {
                    const int idx = slane + it * 64;
                    const int row = idx / CPR, c = idx - row * CPR, rrow = row < 16 ? row : 15;
                    const int m = mw0 + i * 16 + rrow, ncol = nout0 + c * VEC;
                    if (ncol < Nout) {
                        // which CFG half the row belongs to: one division per 16-row slab when the halves are 16-row
                        // aligned (every production shape), per row otherwise
                        const bool odd = p.gate2 && ((((p.rows_per_batch & 15) == 0 ? slab_half_div : m / p.rows_per_batch)) & 1);
                        const float* gsel = odd ? p.gate2 : p.gate;
#pragma unroll
                        for (int e = 0; e < VEC; e += 4) {
                            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gsel + ncol + e);
#pragma unroll
                            for (int f = 0; f < 4; ++f) v[e + f] *= g4[f];
                        }
                    }
                }
                if (has_res) {
                    const V16 r = __builtin_bit_cast(V16, rres[i][it]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += (float)r[e];
                }
                V16 o16;
                if constexpr (sizeof(T) == 2) {             // packed conversions, two values per instruction
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int e = 0; e < VEC; e += 2) {
                        const h16x2 pr = __builtin_convertvector((f32x2){v[e], v[e + 1]}, h16x2);
                        o16[e] = pr[0]; o16[e + 1] = pr[1];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o16[e] = (T)v[e];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o16), rO, (int)store_v(i, it), 0, 0);
            }
        }
        }
        if constexpr (GNS) {
            static_assert(sizeof(T) == 2 && CONV && !GEGLU && 64 % CPR == 0 && (CPR == 8 || CPR == 16), "epilogue statistics: 64- or 128-column wave tiles");
            // fold the lanes that share a chunk column (lane % CPR): fixed tree over the upper lane bits
#pragma unroll
            for (int off = 32; off >= CPR; off >>= 1)
#pragma unroll
                for (int k = 0; k < 4; ++k) gq[k] += __shfl_xor(gq[k], off, 64);
            int glane = lane;
            asm volatile("" : "+v"(glane));
            if (glane < CPR) {
                const int img = em0 / p.gn_hw, chunk = ((em0 - img * p.gn_hw) / BM) * WM + wm_u;       // uniform
                const int quads = p.N >> 2, q0 = ((en0 + wn_u * WBN) >> 2) + glane * 2;
                float* o = p.gn_part + (((size_t)img * (p.gn_hw / (BM / WM)) + chunk) * quads + q0) * 2;
                *reinterpret_cast<f32x4*>(o) = (f32x4){gq[0], gq[2], gq[1], gq[3]};
            }
        }
        STAMP(st_[5]);
        STAMP_ACC(0); STAMP_ACC(1); STAMP_ACC(2); STAMP_ACC(3); STAMP_ACC(4);
        if (!more) break;
        // the staging state was dead during the epilogue (registers!): re-derive it for the K loop.  The opaque
        // copy keeps the compiler from holding the pre-epilogue values live across the epilogue instead.
        vb = nvb;
        tile_par ^= 1;
        asm volatile("" : "+s"(vb));
        setup(vb);
        derive(0);
        b0 = xbuf ^ 1;
        STAMP(st_[6]);
        STAMP_ACC(5);
    }
#if defined(DSIM_DEVTOOLS) && defined(DSIM_STAMPS)
    // phase sums of this wave: [bias / accumulator init | loop-top wait + barrier | K loop | next tile's setup + first stage | epilogue | re-derive]
    if (p.stamps && lane == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(p.stamps + i, sa_[i]);
    if (p.stamps && tid == 0) atomicAdd(p.stamps + 6, 1ull);
#endif
}

}  // namespace

int cu_count() {
    static int per_dev[64] = {0};      // CU count of each device this process has launched on
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = __atomic_load_n(&per_dev[dev], __ATOMIC_RELAXED);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&per_dev[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

// Width (in tile columns) of the column bands the logical tile order walks (gemm_kernel setup()).  An XCD runs ~32 tiles at
// a time; in row-major order at N = 10240 (40 tile columns) those are one activation panel x 32 different weight tiles,
// 21 MB of weights against a 4 MiB L2: every tile re-fetched its whole weight tile over the fabric (7.1 GB per launch
// measured for 1.1 GB algorithmic).  With bands of gn columns the concurrent set is (32 / gn) panels x gn weight tiles
// and the band's weights (gn x BN x K) are re-used from L2 down the whole M sweep; the activations are then read
// tilesN / gn times (2.5-3.2 GB per launch at gn = 3..4: profiles/r04_experiments.txt item 1).  gn is 5 when five weight
// tiles fit in ~2.5 MB of the L2, else 3: the ODD widths measured 2-5 % faster than 4 or 8 at equal or higher traffic (32
// concurrent tiles are then not a whole number of panels, so an XCD's workgroups do not all sit on the same K offset of
// the same lines).  Per-tile arithmetic is untouched: outputs are bit-identical for every gn.
int gemm_band_width(int tilesM, int tilesN, size_t w_tile_bytes) {
    (void)tilesM;
    if (tilesN <= 4) return tilesN;                        // already one band (row-major order)
    int gn = (size_t)5 * w_tile_bytes <= (size_t)2560 * 1024 ? 5 : 3;
    // even out the bands: 12 tile columns at gn = 5 -> 3 bands of 4 ... but keep the width odd: 4 bands of 3
    const int bands = (tilesN + gn - 1) / gn;
    int even = (tilesN + bands - 1) / bands;
    if (!(even & 1)) even = even > 3 ? even - 1 : 3;
    return even < gn ? even : gn;
}

// byte extents of the operands for the buffer descriptors (32-bit offsets: every tensor < 2 GiB)
int gemm_fill_extents(GemmArgs& g, size_t es) {
    size_t a0b, a1b = 16;
    if (g.mode == GEMM_CONV3) {
        const size_t bimg = (size_t)g.M / ((size_t)g.Hout * g.Wout);
        a0b = bimg * g.Hin * g.Win * g.C0 * es;
    } else {
        a0b = (size_t)g.M * g.C0 * es;
        if (g.A1) a1b = (size_t)g.M * g.C1 * es;
    }
    const size_t wb = (size_t)g.N * g.K * es + (g.wb_rows ? (size_t)(g.M / g.wb_rows - 1) * g.wb_stride : 0);
    const size_t ob = (size_t)g.M * g.ldo * es;        // output (and residual) extent: rows are ldo elements apart
    if (a0b >= 0x7fffffffull || a1b >= 0x7fffffffull || wb >= 0x7fffffffull || ob >= 0x7fffffffull) return DSIM_ERR_INVALID;
    g.a0_bytes = (unsigned)a0b; g.a1_bytes = (unsigned)a1b; g.w_bytes = (unsigned)wb; g.out_bytes = (unsigned)ob;
    return DSIM_OK;
}

namespace {

template <typename T, int BM, int BN, int MODE, bool GEGLU, int WM, int WN, int EK>
int launch_ek(const GemmArgs& a, hipStream_t s) {
    constexpr int NW = WM * WN;
    constexpr int LDS = gemm_lds_bytes<T, BM, BN, GEGLU, WM, WN>();
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static DeviceOnce once;          // the LDS opt-in is a per-device attribute of the function
    auto kern = gemm_kernel<T, BM, BN, MODE, GEGLU, WM, WN, EK>;
    CK_ONCE(once, kern, LDS);
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    GemmArgs g = a;
    const size_t es = sizeof(T);
    {
        const int st = gemm_fill_extents(g, es);
        if (st != DSIM_OK) return st;
    }
#ifdef DSIM_DEVTOOLS
    g.exp = g_gemm_exp;
    g.stamps = g_gemm_stamps;
#endif
    // persistent grid: as many workgroups as stay resident (LDS-limited), a multiple of 8 so a workgroup keeps its XCD
    const int ntiles = tilesM * tilesN;
    const int resident = ((cu_count() * (LDS <= 80 * 1024 ? 2 : 1)) / 8) * 8;
    const int grid = ntiles <= resident || resident < 8 || !g_gemm_persistent ? ntiles : resident;
    int gn = gemm_band_width(tilesM, tilesN, (size_t)BN * a.K * es);
#ifdef DSIM_DEVTOOLS
    if (g_gemm_exp >> 16) gn = std::min(tilesN, g_gemm_exp >> 16);      // kbench: KB_GEXP = gn << 16 (>= tilesN: row-major order)
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, s, g, tilesN, ntiles, tilesM, gn);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

template <typename T, int BM, int BN, int MODE, bool GEGLU, int WM = 4, int WN = 1, bool SLOW = false>
int launch_one(const GemmArgs& a, hipStream_t s) {
    if constexpr (SLOW) {
        if (a.act == 1 && !a.gate && a.epi != EPI_RESIDUAL) return launch_ek<T, BM, BN, MODE, GEGLU, WM, WN, EK_ACT>(a, s);
        return launch_ek<T, BM, BN, MODE, GEGLU, WM, WN, EK_SLOW>(a, s);
    }
    if constexpr (!GEGLU && sizeof(T) == 2 && MODE == GEMM_CONV3P && ((BM == 256 && (BN == 128 || BN == 256)) || (BM == 512 && BN == 128))) {
        // GroupNorm statistics from the epilogue (GemmArgs.gn_part): the VAE's 512 x 512 / 256 x 256 levels on their 256-row tiles
        if (a.gn_part) {
            if (a.gn_hw <= 0 || a.gn_hw % BM || a.M % a.gn_hw || a.N % BN || a.bias2) return DSIM_ERR_INVALID;
            return a.epi == EPI_RESIDUAL ? launch_ek<T, BM, BN, MODE, GEGLU, WM, WN, EK_RES_GN>(a, s)
                                         : launch_ek<T, BM, BN, MODE, GEGLU, WM, WN, EK_PLAIN_GN>(a, s);
        }
    }
    if (a.gn_part) return DSIM_ERR_INVALID;            // asked for on a tile that has no statistics epilogue (gemm_gn_stats_tile() says which)
    if constexpr (!GEGLU)
        if (a.epi == EPI_RESIDUAL) return launch_ek<T, BM, BN, MODE, GEGLU, WM, WN, EK_RES>(a, s);
    return launch_ek<T, BM, BN, MODE, GEGLU, WM, WN, EK_PLAIN>(a, s);
}

int check_args(const GemmArgs& a, int BK) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % BK) return DSIM_ERR_INVALID;
    if (a.mode == GEMM_CONV3) {
        if (a.C0 % BK || a.K != 9 * a.C0 || a.A1) return DSIM_ERR_INVALID;
    } else {
        if (a.C0 % BK || (a.A1 && a.C1 % BK) || a.K != a.C0 + (a.A1 ? a.C1 : 0)) return DSIM_ERR_INVALID;
    }
    if (a.epi == EPI_GEGLU && (a.mode != GEMM_LINEAR || (a.geglu_blk == 16 ? a.N % 320 != 0 : (a.geglu_blk != 32 || a.N % 64))))
        return DSIM_ERR_INVALID;
    if (!a.zero_page) return DSIM_ERR_INVALID;
    if (a.wb_rows && (a.wb_rows < 0 || a.mode != GEMM_LINEAR || a.M % a.wb_rows || a.A1 || a.epi == EPI_GEGLU || a.out_split ||
                      a.wb_stride % 16))
        return DSIM_ERR_INVALID;
    if (a.mode == GEMM_CONV3 && a.epi == EPI_RESIDUAL && a.bias2) return DSIM_ERR_INVALID;      // per-half bias: plain convs only (conv1 of a resnet)
    if (a.out_split && (a.mode != GEMM_LINEAR || a.epi != EPI_NONE || a.act || a.gate || a.out_split % 320 || a.N % a.out_split ||
                        a.out_split_stride <= 0 || (a.N / a.out_split) * a.out_split_stride >= 0x7fffffffll))
        return DSIM_ERR_INVALID;
    return DSIM_OK;
}

template <typename T>
int launch_typed(const GemmArgs& a_in, hipStream_t s) {
    const int st = check_args(a_in, Traits<T>::BK);
    if (st != DSIM_OK) return st;
    GemmArgs a = a_in;
    a.lwo = a.lhw = -1;
#ifdef DSIM_DEVTOOLS
    if (a.epi == EPI_GEGLU && (g_gemm_exp & 8192) && a.N % 64 == 0) a.geglu_blk = 32;      // kbench: the 256 / 128-column GEGLU tiles
#endif
    if (a.mode == GEMM_CONV3) {          // power-of-two output maps: the kernels split the pixel index with shifts
        const int hw = a.Hout * a.Wout;
        if (a.Wout > 0 && !(a.Wout & (a.Wout - 1)) && !(hw & (hw - 1))) {
            a.lwo = a.lhw = 0;
            while ((1 << a.lwo) < a.Wout) ++a.lwo;
            while ((1 << a.lhw) < hw) ++a.lhw;
        }
    }
    if (a.wb_rows == a.M) a.wb_rows = 0;        // one batch: a plain GEMM
    if constexpr (sizeof(T) == 2) {
        // problems too small to fill the chip: 64 x 64 tiles behind a deep LDS ring, the same arithmetic bit for bit (gemm_skinny.hip)
        if (g_gemm_skinny && !a.gn_part && !a.wb_rows && !a.force_big && gemm_skinny_applies(a)) {
            GemmArgs g = a;
            const int se = gemm_fill_extents(g, sizeof(T));
            return se != DSIM_OK ? se : launch_gemm_skinny(g, s);
        }
    }
    const bool slow = a.act != 0 || a.gate != nullptr;     // DiT linears only
    if (slow && (a.mode != GEMM_LINEAR || a.epi == EPI_GEGLU)) return DSIM_ERR_INVALID;
    int bm, bn;
    gemm_launch_tile(a, sizeof(T) == 2 ? DSIM_H16 : DSIM_F32, &bm, &bn);
    // out_split: the epilogue picks ONE destination tensor per tile from its first column, so the tile width must divide the split
    if (a.out_split && a.out_split % bn != 0) return DSIM_ERR_INVALID;
    if (a.wb_rows % bm != 0) return DSIM_ERR_INVALID;           // a tile's rows belong to one weight batch
    if constexpr (sizeof(T) == 2)
        if (bm == 512) return a.lwo >= 0 && bn == 128 ? launch_one<T, 512, 128, GEMM_CONV3P, false, 8, 1>(a, s) : DSIM_ERR_INVALID;
    const bool big = bm == 256, n160 = bn == 160;
    (void)big;
    if constexpr (sizeof(T) == 2) {
        // h16, big problems: 256-row tiles, 8 waves as 4(M) x 2(N), 64-row x (BN/2)-column sub-tiles
        if (big) {
            if (a.epi == EPI_GEGLU)
                return bn == 320 ? launch_one<T, 256, 320, GEMM_LINEAR, true, 4, 2>(a, s) : launch_one<T, 256, 256, GEMM_LINEAR, true, 4, 2>(a, s);
            if (a.mode == GEMM_CONV3) {
                // (the VAE's 128-channel levels: power-of-two maps as well -- setup()'s integer divisions were 10 % of these K = 1152 tiles)
                if (bn == 128) return a.lwo >= 0 ? launch_one<T, 256, 128, GEMM_CONV3P, false, 4, 2>(a, s)
                                                 : launch_one<T, 256, 128, GEMM_CONV3, false, 4, 2>(a, s);
                // power-of-two output maps (every SD level at the sizes these tiles serve): the instantiation without the integer
                // divisions in setup(); a run-time branch instead spilled scalar registers in the residual kernel
                if (a.lwo >= 0) return bn == 320 ? launch_one<T, 256, 320, GEMM_CONV3P, false, 4, 2>(a, s)
                                                 : launch_one<T, 256, 256, GEMM_CONV3P, false, 4, 2>(a, s);
                return bn == 320 ? launch_one<T, 256, 320, GEMM_CONV3, false, 4, 2>(a, s)
                                 : launch_one<T, 256, 256, GEMM_CONV3, false, 4, 2>(a, s);
            }
            if (slow && bn == 320) return launch_ek<T, 256, 320, GEMM_LINEAR, false, 4, 2, EK_ACT>(a, s);      // tanh-GELU only (DiT Mlp.fc1)
            if (slow) return bn == 192 ? launch_one<T, 256, 192, GEMM_LINEAR, false, 4, 2, true>(a, s)
                                       : launch_one<T, 256, 256, GEMM_LINEAR, false, 4, 2, true>(a, s);
            if (bn == 192) return launch_one<T, 256, 192, GEMM_LINEAR, false, 4, 2>(a, s);
            return bn == 320 ? launch_one<T, 256, 320, GEMM_LINEAR, false, 4, 2>(a, s)
                             : launch_one<T, 256, 256, GEMM_LINEAR, false, 4, 2>(a, s);
        }
    }
    if (slow) return launch_one<T, 128, 128, GEMM_LINEAR, false, 4, 1, true>(a, s);
    if (a.epi == EPI_GEGLU) return n160 ? launch_one<T, 128, 160, GEMM_LINEAR, true>(a, s) : launch_one<T, 128, 128, GEMM_LINEAR, true>(a, s);
    if constexpr (sizeof(T) == 2)
        if (bn == 80) return a.mode == GEMM_CONV3 ? launch_one<T, 128, 80, GEMM_CONV3, false>(a, s) : launch_one<T, 128, 80, GEMM_LINEAR, false>(a, s);
    if (a.mode == GEMM_CONV3)
        return n160 ? launch_one<T, 128, 160, GEMM_CONV3, false>(a, s) : launch_one<T, 128, 128, GEMM_CONV3, false>(a, s);
    return n160 ? launch_one<T, 128, 160, GEMM_LINEAR, false>(a, s) : launch_one<T, 128, 128, GEMM_LINEAR, false>(a, s);
}

}  // namespace

int launch_gemm(const GemmArgs& a, int dtype, hipStream_t s) {
    if (dtype == DSIM_H16) return launch_typed<h16>(a, s);
#ifndef DSIM_H16_IS_F16            // the fp16 objects hold the fp16 kernels only; the plain names forward DSIM_F16 to them
    if (dtype == DSIM_F32) return launch_typed<float>(a, s);
#ifdef DSIM_HAS_F16_TWINS
    if (dtype == DSIM_F16) return launch_gemm_f16(a, dtype, s);
#endif
#endif
    return DSIM_ERR_INVALID;
}

}  // namespace dsim
