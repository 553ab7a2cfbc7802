// The 256-token, d = 160 attention core of SD1.5's 16 x 16 level for gfx950 -- the shape BASELINE.json's utilisation target is
// stated on ("up_blocks[0] attention GEMM"):
//
//  * pair_tail160_kernel : the DiffSim score tail at the default tap -- /root/reference/diffsim/diffsim.py:177-197:
//                          O_aa = SDPA(Qa, Ka, Va), O_ab = SDPA(Qa, Kb, Vb) (and the b <-> a mirror), cosine / mse over the
//                          flattened (B, H, N, D) tensors -- for N = 256 tokens, head dim 160, 16-bit compute types.
//  * sdpa160_kernel      : the same core as a plain SDPA -- the U-Net's own self-attentions of that level
//                          (/root/reference/diffsim/hacked_attn.py:74-81); described where it is defined.
//
// Why a kernel of its own.  pair_tail_kernel<h16, 160> (attention.hip) gives a 128-query workgroup its own single-buffered
// copy of every key tile: 50 % of its wave cycles wait for global loads or barriers and the matrix pipe is 0.2 busy.  The work
// is MFMA-bound on paper (per 32 x 32 block of S 20 MFMAs = 640 matrix cycles against ~80 vector instructions), so the
// structure here is built around keeping K / V arriving while the MFMAs run:
//
//   - ONE persistent 512-thread workgroup per CU walks "units" = (pair, direction, CFG half, head): wave w owns query rows
//     32 w .. 32 w + 31 of the unit's 256, so the 256 keys of an attention are staged ONCE for all of its queries.
//   - EVERYTHING arrives by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave instruction) issued from inline asm, so that
//     the compiler's own s_waitcnt bookkeeping never sees a vector-memory load in the loop and never drains the stream:
//     K and V as 32-key tiles (10 KB + 10 KB) through a 4-slot ring, three to four tiles ahead, one s_barrier per tile behind a
//     COUNTED vmcnt; the NEXT unit's Q rows into a wave-private 10 KB slab, one piece per tile step.  The stream does not
//     stop at unit borders: the last three steps of a unit issue the first three tiles of the next.
//   - dense 320-byte rows in LDS, no padding: K and Q chunks are XOR-swizzled in their low two bits by (row >> 2) & 3 (applied by
//     the DMA's per-lane source address, so the image the ds_read_b128 fragment reads see is conflict-free), V rows are plain
//     (the transposed ds_read_b64_tr_b16 reads are conflict-free at 80 dwords per row).  4 x 20 KB + 8 x 10 KB = all 160 KB.
//   - swapped QK^T (S^T = K Q^T, a lane owns one query column), P straight from the accumulators into the PV MFMAs' B operand,
//     exact online softmax (the rescale runs only in blocks where some row's maximum grew).  Q is NOT pre-scaled: the softmax
//     computes exp2(fma(s, c, -m c)) with c = log2(e) / sqrt(160), one v_fma per element where the old form spent a subtract.
//   - the self-attention's normalised output, rounded to the compute dtype (what torch's SDPA returns), waits for the cross pass
//     as packed pairs (40 registers); the products run as v_dot2c on the packed pairs, f32 per wave, f64 in a fixed order
//     across waves and units (pair_finish160_kernel): bit-reproducible, and independent of the batch a pair is scored in.
//
// Units are dealt so that the two directions of a (pair, CFG half, head) -- which read the same four K / V tensors -- run at
// the same time on two workgroups of one XCD (blockIdx.x % 8 names the XCD under round-robin placement: speed only).
#include "common.h"

namespace dsim {
namespace {

constexpr int A_D = 160, A_N = 256, A_KT = 32;
constexpr int A_ROWB = A_D * 2;                 // bytes per (row, head) in the 16-bit types
constexpr int A_KTILE = A_KT * A_ROWB;          // 10240: one K (or V) tile image, also one wave's Q slab
constexpr int A_SLOT = 2 * A_KTILE;             // K tile + V tile
constexpr int A_NSLOT = 4;
constexpr int A_PRE = 3;                        // fragment reads in flight ahead of the MFMA that consumes them
constexpr int A_RING = A_NSLOT * A_SLOT;        // 81920
constexpr int A_LDS = A_RING + 8 * A_KTILE;     // 163840 bytes: all of a CU's LDS, one workgroup per CU
constexpr int A_NKS = A_D / 16;                 // 10 k steps over d (QK^T)
constexpr int A_NDB = A_D / 32;                 // 5 output blocks over d (PV)

#ifdef DSIM_H16_IS_F16
#define H16_DOT2(a, b, c) __builtin_amdgcn_fdot2((a), (b), (c), false)
#else
#define H16_DOT2(a, b, c) __builtin_amdgcn_fdot2_f32_bf16((a), (b), (c), false)
#endif

__device__ __forceinline__ float max_halves160(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Maximum of the 16 scores a lane holds after a QK^T chain, as ONE asm statement of v_max3_f32 (fmaxf on MFMA outputs draws a
// canonicalising v_max(x, x) per operand from hipcc: 16 more vector instructions per step).  The statement OPENS with the 12 wait
// states an 8-pass MFMA's result needs before a vector instruction may read it: hipcc pads that hazard for its own instructions
// only, never inside or in front of an asm statement (cdna_hip_programming.md 5.7 item 2) -- without them the maximum was taken
// over whatever the registers held before, i.e. partly over the PREVIOUS tile's scores on some launches: a different softmax
// reference point, hence scores that moved in the seventh digit from run to run and self pairs scoring 0.999997.
__device__ __forceinline__ float max16_after_mfma(const f32x16& s) {
    float r, t1, t2, t3, t4;
    asm volatile("s_nop 11\n\t"
                 "v_max3_f32 %0, %5, %6, %7\n\t"
                 "v_max3_f32 %1, %8, %9, %10\n\t"
                 "v_max3_f32 %2, %11, %12, %13\n\t"
                 "v_max3_f32 %3, %14, %15, %16\n\t"
                 "v_max3_f32 %4, %17, %18, %19\n\t"
                 "v_max3_f32 %0, %0, %1, %2\n\t"
                 "v_max3_f32 %3, %3, %4, %20\n\t"
                 "v_max_f32 %0, %0, %3"
                 : "=&v"(r), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
                 : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]), "v"(s[10]),
                   "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]));
    return r;
}
constexpr float A_THR = 8.0f;                   // log2 units a row maximum may grow past the softmax reference before a rescale

// x + (the value in the lane 32 away): one v_permlane32_swap instead of a ds_bpermute round trip
__device__ __forceinline__ float half_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the 64 lanes (wave-uniform result): DPP row shifts leave every 16-lane row's sum in its last lane, four readlanes add
// the rows -- VALU only (the __shfl_xor tree is six dependent LDS round trips per value), in a fixed order
__device__ __forceinline__ float wave_sum(float x) {
    int v = __float_as_int(x);
#define DSIM_DPP_ADD(ctrl) v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false)))
    DSIM_DPP_ADD(0x111);            // row_shr:1
    DSIM_DPP_ADD(0x112);            // row_shr:2
    DSIM_DPP_ADD(0x114);            // row_shr:4
    DSIM_DPP_ADD(0x118);            // row_shr:8   -> lane 15 of every row holds the row's sum
#undef DSIM_DPP_ADD
    return (__int_as_float(__builtin_amdgcn_readlane(v, 15)) + __int_as_float(__builtin_amdgcn_readlane(v, 31))) +
           (__int_as_float(__builtin_amdgcn_readlane(v, 47)) + __int_as_float(__builtin_amdgcn_readlane(v, 63)));
}

template <int N> struct IC { static constexpr int value = N; };

// In-kernel phase stamps (-DDSIM_DEVTOOLS -DDSIM_STAMPS builds of tools/kbench only; the s_waitcnt behind each s_memtime also
// drains the wave's LDS reads, so a stamped build is slower and less pipelined than the product -- it says where time goes, not how much)
#if defined(DSIM_DEVTOOLS) && defined(DSIM_STAMPS)
#define TSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tst_[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TSTAMP_ACC(i) tsa_[i] += tst_[(i) + 1] - tst_[i]
#else
#define TSTAMP(i) do { } while (0)
#define TSTAMP_ACC(i) do { } while (0)
#endif

// s_waitcnt vmcnt(N) for a compile-time N (the ring's counted waits)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// One LDS-DMA piece: lane i's 16 bytes at desc.base + voff + soff land at LDS byte lds + 16 i.  Inline asm: hipcc neither counts it
// in its s_waitcnt bookkeeping nor orders LDS reads behind it -- the caller's counted vmcnt + barrier do (cdna_hip_programming.md 5.7).
// M0 is saved and restored inside the statement; the two s_mov + s_nop 2 are also the five wait states a VALU-written SGPR operand
// (hipcc parks scalars in VGPR lanes under SGPR pressure: v_readlane right in front of the statement) needs before a VMEM reads it.
__device__ __forceinline__ void dma_piece(unsigned lds, int voff, const u32x4& desc, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 make_desc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 d = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return d;
}

// a unit's tensors as BYTE offsets into the three allocations (q, k and v have one layout, so two offsets describe all five views:
// four scalar registers per unit instead of ten; the descriptors are formed from them at the point of issue)
struct Unit160 {
    unsigned long long oq;              // the query image's rows
    unsigned long long o0, o1;          // the K / V of the pair's FIRST image (pass 0: steps 0-7) and of its SECOND image (steps 8-15) --
                                        // in BOTH directions, so that the two workgroups of a couple stream the same tensors at the same
                                        // time (direction 1 runs its cross pass first)
    int pidx;                           // this wave's slot in the partial array
};

__global__ __launch_bounds__(512, 2) void pair_tail160_kernel(const h16* __restrict__ qg, const h16* __restrict__ kg,
                                                              const h16* __restrict__ vg, const int32_t* __restrict__ idx_a,
                                                              const int32_t* __restrict__ idx_b, const int n_pairs, const int B,
                                                              const int H, const float c, const int mse,
                                                              float* __restrict__ part, char* __restrict__ park
#ifdef DSIM_DEVTOOLS
                                                              , float* __restrict__ dbg, const int exp
#endif
                                                              ) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int ld = H * A_D;
    const int rowb = ld * 2;
    const size_t img = (size_t)B * A_N * ld;
    const int BH = B * H;
    const int total = n_pairs * BH;                 // (pair, CFG half, head) items; each is two units (directions)
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, cj = wslot >> 1, dir = wslot & 1, npc = (int)gridDim.x >> 4;
    const unsigned recs = (unsigned)((A_N - 1) * rowb + A_ROWB);    // bytes of one (image, CFG half, head) view from its first element

    // ---- DMA pieces.  A piece is 1 KB = 3.2 dense rows of a 32-row image: lane i writes LDS byte 1024 p + 16 i and chooses its
    // source chunk (K, Q: swizzled).  The lane pattern repeats every 5 pieces = 16 rows.
    auto piece_voff = [&](int pat, bool swizzle, int ln, int rb) {
        const int f = 64 * pat + ln;
        const int r = (f * 3277) >> 16, pos = f - r * 20;            // f / 20, f % 20 (f < 320)
        const int ch = swizzle ? ((pos & ~3) | ((pos & 3) ^ (r >> 2))) : pos;
        return r * rb + ch * 16;
    };
    // K / V: this wave's pieces of a tile are j = wave, wave + 8, wave + 16 (< 20) of [K pieces 0..9 | V pieces 0..9]: waves 0-3 issue
    // three, waves 4-7 two (the two waves of a SIMD five together)
    int kv_voff[3], kv_grp[3];
    unsigned kv_lds[3];
    bool kv_isv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pj = wave + 8 * j;
        kv_isv[j] = pj >= 10;
        const int p = kv_isv[j] ? pj - 10 : pj;
        const int grp = p >= 5, pat = p - 5 * grp;
        kv_voff[j] = piece_voff(pat, !kv_isv[j], lane, rowb);
        kv_grp[j] = grp;
        kv_lds[j] = (kv_isv[j] ? A_KTILE : 0) + p * 1024;
    }
#ifdef DSIM_DEVTOOLS
#define T160_ABL(bit) (exp & (bit))
#else
#define T160_ABL(bit) false
#endif
    const bool three = wave < 4;
#ifdef DSIM_DEVTOOLS
    if (exp & 512) { kv_voff[0] = lane * 16; kv_voff[1] = lane * 16 + 1024; kv_voff[2] = lane * 16 + 2048; }
    {   // kbench: static wave priority (exp bits 0-1: level; bit 2: for waves 4-7 instead of 0-3)
        const bool mine = (exp & 4) ? !three : three;
        if (mine) {
            if ((exp & 3) == 1) __builtin_amdgcn_s_setprio(1);
            else if ((exp & 3) == 2) __builtin_amdgcn_s_setprio(2);
            else if ((exp & 3) == 3) __builtin_amdgcn_s_setprio(3);
        }
    }
#endif
    auto issue_kv = [&](unsigned long long uoff, int tile, int slot) {
        if (T160_ABL(16)) return;            // kbench ablation: no DMA after the prologue's (timing only)
        const u32x4 dK = make_desc((const char*)kg + uoff, recs), dV = make_desc((const char*)vg + uoff, recs);
        // (scalars re-derived per hand-over from an opaque copy: hipcc otherwise hoists ~60 loop-invariant offsets out of the unit loop,
        // parks them in VGPR lanes and pays a v_readlane + hazard padding for each in front of the DMA that uses it)
        int rb = rowb;
        unsigned lb = lbase;
        asm volatile("" : "+s"(rb), "+s"(lb));
        const unsigned sb = lb + slot * A_SLOT;
        const int ts = tile * A_KT * rb;
        dma_piece(sb + kv_lds[0], kv_voff[0], dK, ts + kv_grp[0] * 16 * rb);                        // j = 0: always a K piece
        dma_piece(sb + kv_lds[1], kv_voff[1], kv_isv[1] ? dV : dK, ts + kv_grp[1] * 16 * rb);
        if (three) dma_piece(sb + kv_lds[2], kv_voff[2], dV, ts + kv_grp[2] * 16 * rb);             // j = 2: always a V piece
    };
    // Q: piece j (0..9) of this wave's 32 rows into its slab
    const unsigned qslab = lbase + A_RING + wave * A_KTILE;
    auto issue_q = [&](unsigned long long uoff, int j) {
        if (T160_ABL(16)) return;
        const u32x4 dQ = make_desc((const char*)qg + uoff, recs);
        int ln = lane, rb = rowb;
        unsigned qs = qslab;
        asm volatile("" : "+v"(ln), "+s"(rb), "+s"(qs));     // recomputed per piece (see issue_kv; five hoisted lane patterns would be spilled too)
        dma_piece(qs + j * 1024, piece_voff(j % 5, true, ln, rb), dQ, (wave * 32 + 16 * (j / 5)) * rb);
    };

    // ---- fragment read addresses (bytes inside a 32-row image / a slot) ------------------------------------------------------
    const int swz = (l31 >> 2) & 3;
    const int e0 = l31 * A_ROWB + ((half ^ swz) << 4);              // even k steps: chunk 4 (ks >> 1) + (half ^ swz)
    const int e1 = l31 * A_ROWB + (((2 + half) ^ swz) << 4);        // odd k steps:  chunk 4 (ks >> 1) + ((2 + half) ^ swz)
    const int vl = A_KTILE + (4 * half + ((lane & 15) >> 2)) * A_ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    auto setup = [&](int p2) {
        Unit160 u;
        const int pair = p2 / BH, bh = p2 - pair * BH, b = bh / H, h = bh - b * H;
        const int ia = __builtin_amdgcn_readfirstlane(idx_a[pair]), ib = __builtin_amdgcn_readfirstlane(idx_b[pair]);
        const int iq = dir ? ib : ia;
        const unsigned long long off = ((unsigned long long)b * A_N * ld + h * A_D) * 2ull;
        u.oq = (unsigned long long)iq * (img * 2) + off;
        u.o0 = (unsigned long long)ia * (img * 2) + off;
        u.o1 = (unsigned long long)ib * (img * 2) + off;
        u.pidx = ((pair * 2 + dir) * BH + bh) * 8 + wave;
        return u;
    };
    h16x8 q[A_NKS];
    auto read_q = [&]() {                            // this wave's Q fragments from its slab
        const char* s0 = smem + A_RING + wave * A_KTILE;
#pragma unroll
        for (int ks = 0; ks < A_NKS; ++ks) q[ks] = *reinterpret_cast<const h16x8*>(s0 + ((ks & 1) ? e1 : e0) + (ks >> 1) * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // in registers before the slab is refilled
    };

    int it = 0;
    const int p20 = cj * 8 + xcd;
    if (p20 >= total) return;
    Unit160 cur = setup(p20);
#pragma unroll
    for (int j = 0; j < 10; ++j) issue_q(cur.oq, j);
#pragma unroll
    for (int t = 0; t < A_NSLOT - 1; ++t) issue_kv(cur.o0, t, t);
    if (three) issue_kv(cur.o0, A_NSLOT - 1, A_NSLOT - 1);        // (waves 4-7 issue it in step 0: see the hand-over)
    if (three) wait_vm<3 * A_NSLOT>(); else wait_vm<2 * (A_NSLOT - 1)>();
    read_q();
    // tile 0 visible to every wave; its first K fragments
    if (three) wait_vm<3 * (A_NSLOT - 1)>(); else wait_vm<2 * (A_NSLOT - 2)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    h16x8 kpre[A_PRE];
#pragma unroll
    for (int ks = 0; ks < A_PRE; ++ks) kpre[ks] = *reinterpret_cast<const h16x8*>(smem + ((ks & 1) ? e1 : e0) + (ks >> 1) * 64);

    f32x16 o[A_NDB];
    float m_run = 0.f, l_run = 0.f;
    const float thr = A_THR / c;                    // the threshold in raw-logit units
    // the self pass's output waits for the cross pass in a 10 KB slab per wave of the workspace (L2-resident: written and read
    // back by the same CU a few microseconds apart) instead of 40 registers held through eight steps
    const u32x4 dP = make_desc(park + ((size_t)blockIdx.x * 8 + wave) * A_KTILE, A_KTILE);

    // the pass's output, normalised and rounded to the compute dtype, two values per register (d = 32 db + (r & 3) + 8 (r >> 2) + 4 half)
    auto pack_o = [&](u32x4 (&pk)[2 * A_NDB]) {
        const float inv = __builtin_amdgcn_rcpf(half_sum(l_run));      // (v_rcp_f32: 1 ulp, against the 8 bits the outputs keep)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int db = 0; db < A_NDB; ++db)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const h16x2 v = __builtin_convertvector((f32x2){o[db][2 * i] * inv, o[db][2 * i + 1] * inv}, h16x2);
                pk[2 * db + (i >> 2)][i & 3] = __builtin_bit_cast(unsigned, v);
            }
    };

#if defined(DSIM_DEVTOOLS) && defined(DSIM_STAMPS)
    unsigned long long tst_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tsa_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        const int p2n = ((it + 1) * npc + cj) * 8 + xcd;
        const bool has_next = p2n < total;
        // (no next unit: the look-ahead re-fetches this unit's own tiles and rows -- same instruction counts, no special cases)
        const Unit160 nxt = has_next ? setup(p2n) : cur;

        // One 32-key tile; steps 0-7 are the self pass, 8-15 the cross pass.  A step is  QK^T(T) . softmax(T) . PV(T), and the
        // hand-over to tile T + 1 sits INSIDE the PV: once the step's last V^T fragments are in registers (three MFMAs before its
        // end) the wave waits for its own DMA pieces of tile T + 1, meets the others at the barrier, issues the DMA of tile T + 4
        // into the slot tile T just left and the first K fragment reads of tile T + 1 -- and runs the remaining three MFMAs from
        // registers while those reads are in flight, instead of idling on LDS latency behind every barrier.
        // Vector-memory operations per hand-over in issue order: in steps 0-9 one Q piece of the next unit, then the K / V pieces of
        // tile T + 4 (3 or 2 per wave).  The hand-over of step T needs the pieces issued three hand-overs earlier.
        u32x4 ypk[2 * A_NDB];
        auto step = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            const bool idle = (T160_ABL(128) && !three) || (T160_ABL(256) && three);          // kbench ablation 128: waves 4-7 only stage (one computing wave per SIMD)
            constexpr bool FIRST = (T & 7) == 0;
            const char* sb = smem + (T & 3) * A_SLOT;
            TSTAMP(0);
            // ---- S^T = K Q^T: keys x this lane's query column, raw (unscaled) logits; fragments A_PRE reads ahead -------------
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            h16x8 kf[A_PRE + 1];
#pragma unroll
            for (int i = 0; i < A_PRE; ++i) kf[i] = kpre[i];
            if (!idle) {
#pragma unroll
            for (int ks = 0; ks < A_NKS; ++ks) {
                if (ks + A_PRE < A_NKS)
                    kf[(ks + A_PRE) % (A_PRE + 1)] = *reinterpret_cast<const h16x8*>(sb + (((ks + A_PRE) & 1) ? e1 : e0) + ((ks + A_PRE) >> 1) * 64);
                __builtin_amdgcn_sched_barrier(0);
                s = H16_MFMA_32x32x16(kf[ks % (A_PRE + 1)], q[ks], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            TSTAMP(1);
            if constexpr (T == 13) {
                // The parked self output comes back two steps before it is used: older than the DMA pieces of hand-overs 13 to 15, it has
                // landed by the time hand-over 15's own wait has passed, and the epilogue waits for nothing.  Inline asm, so that hipcc
                // does not wait for these loads with a count that ignores the DMA pieces behind them.  sc1: served by L2 -- this CU's L1 may still hold
                // the slab's lines from the previous unit's read-back.  s_nop 4: an SGPR operand hipcc has just restored from a VGPR lane
                // (v_readlane) needs five wait states before a VMEM reads it, and hipcc pads nothing inside an asm statement.
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int po = ln * 16;
#pragma unroll
                for (int i = 0; i < 2 * A_NDB; ++i)
                    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=v"(ypk[i]) : "v"(po), "s"(dP), "s"(i * 1024) : "memory");
            }
            // the first V^T fragments: their LDS latency passes under the softmax
            const char* vb = sb + vl;
            auto vread = [&](int j) {                   // fragment of PV MFMA j = 5 s2 + db
                const char* pa = vb + (j / A_NDB) * 16 * A_ROWB + (j % A_NDB) * 64;
                const h16x4 lo = h16_ds_read_tr16_b64(pa);
                const h16x4 hi = h16_ds_read_tr16_b64(pa + 8 * A_ROWB);
                h16x8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                return vf;
            };
            h16x8 vf[A_PRE + 1];
#pragma unroll
            for (int j = 0; j < A_PRE; ++j) vf[j] = vread(j);
            __builtin_amdgcn_sched_barrier(0);
            // ---- online softmax -------------------------------------------------------------------------------------------
            // The reference point m_run moves only when some row's maximum has grown by more than A_THR (log2 units): softmax is
            // invariant to it, P <= 2^A_THR keeps its relative precision in either 16-bit type, and the sums are f32.  (With the
            // exact rule the rescale -- 80 multiplies against the step's 20 MFMAs -- ran in nearly every step: one of 32 rows
            // almost always finds a new maximum among 32 more keys.)
            float tmax = max16_after_mfma(s);
            tmax = max_halves160(tmax);
            if constexpr (FIRST) {
                m_run = tmax;
            } else {
                if (!__all(tmax <= m_run + thr)) {
                    const float mn = fmaxf(m_run, tmax);
                    const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * c);
                    m_run = mn;
                    l_run *= alpha;
#pragma unroll
                    for (int db = 0; db < A_NDB; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                }
            }
            const float mc = -m_run * c;
            if (!T160_ABL(8) && !idle) {            // (kbench ablation 8: no exponentials)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, mc));
            }
            const float psum = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) +
                               (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
            l_run = FIRST ? psum : l_run + psum;
            h16x8 pf[2];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[f][e] = (h16)s[8 * f + e];
            // ---- O^T += V^T P^T, and the hand-over to tile T + 1 ---------------------------------------------------------------
            constexpr int NPV = 2 * A_NDB;
            TSTAMP(2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NPV; ++j) {
                if (j + A_PRE < NPV) vf[(j + A_PRE) % (A_PRE + 1)] = vread(j + A_PRE);
                if (j == NPV - A_PRE) {
                    // Waves 0-3 issue their DMA pieces right BEHIND the barrier (below), waves 4-7 -- their SIMD partners -- just in
                    // FRONT of the next one (here: the pieces of the previous hand-over).  Same work, but the two waves of a SIMD
                    // now run half a step apart: one is in its issue / softmax phase while the other has the matrix pipe.
                    if (!three) {
                        constexpr int TL = T + A_NSLOT - 1;
                        if constexpr (T >= 1 && T <= 10) issue_q(nxt.oq, T - 1);
                        if constexpr (TL < 16) issue_kv(TL < 8 ? cur.o0 : cur.o1, TL & 7, TL & 3);
                        else issue_kv(nxt.o0, TL - 16, TL & 3);
                    }
                    // every LDS read of tile T has been issued (and, behind this wait, has returned): its slot may be refilled
                    TSTAMP(3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // issued since the pieces this hand-over needs: the K / V pieces of two hand-overs (6 or 4), the Q pieces of hand-overs
                    // T - 2 and T - 1, the ten park stores behind step 7 (waves 4-7 issue the pieces of hand-over 7 after them, in step 8),
                    // the ten park loads of step 13 (waves 4-7 issue the pieces of hand-over 12 after them)
                    constexpr int QP = (T >= 2 && T <= 11 ? 1 : 0) + (T >= 1 && T <= 10 ? 1 : 0) + (T == 13 || T == 14 ? 10 : 0);
                    if (three) wait_vm<6 + QP + (T >= 8 && T <= 10 ? 10 : 0) + (T == 15 ? 10 : 0)>(); else wait_vm<4 + QP + (T >= 8 && T <= 9 ? 10 : 0)>();
                    TSTAMP(4);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    TSTAMP(5);
                    if (three) {
                        constexpr int TN = T + A_NSLOT;
                        if constexpr (T < 10) issue_q(nxt.oq, T);
                        if constexpr (TN < 16) issue_kv(TN < 8 ? cur.o0 : cur.o1, TN & 7, TN & 3);
                        else issue_kv(nxt.o0, TN - 16, TN & 3);
                    }
                    const char* sn = smem + ((T + 1) & 3) * A_SLOT;
#pragma unroll
                    for (int ks = 0; ks < A_PRE; ++ks) kpre[ks] = *reinterpret_cast<const h16x8*>(sn + ((ks & 1) ? e1 : e0) + (ks >> 1) * 64);
                    TSTAMP(6);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int s2 = j / A_NDB, db = j % A_NDB;
                if (idle) {
                } else if (FIRST && s2 == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    o[db] = H16_MFMA_32x32x16(vf[j % (A_PRE + 1)], pf[s2], z, 0, 0, 0);
                } else {
                    o[db] = H16_MFMA_32x32x16(vf[j % (A_PRE + 1)], pf[s2], o[db], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TSTAMP(7);
            TSTAMP_ACC(0); TSTAMP_ACC(1); TSTAMP_ACC(2); TSTAMP_ACC(3); TSTAMP_ACC(4); TSTAMP_ACC(5); TSTAMP_ACC(6);
            if constexpr (T == 7) if (!T160_ABL(64)) {
                u32x4 ysv[2 * A_NDB];
                pack_o(ysv);
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(park + ((size_t)blockIdx.x * 8 + wave) * A_KTILE), 0, A_KTILE, 0x00020000);
#pragma unroll
                // (the slab offset rides in the VECTOR offset: a buffer_store_dwordx4 with an SGPR soffset reads its data registers late and
                // hipcc pads only for a constant offset: profiles/r05_experiments.txt item 3b)
                for (int i = 0; i < 2 * A_NDB; ++i) __builtin_amdgcn_raw_buffer_store_b128(ysv[i], rP, ln * 16 + i * 1024, 0, 0);
            }
#if defined(DSIM_DEVTOOLS) && !defined(DSIM_STAMPS)
            if constexpr (T == 7 || T == 15) {           // kbench: the first unit's two outputs, f32, [pass][query][d]
                if (dbg && blockIdx.x == 0 && it == 0) {
                    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
#pragma unroll
                    for (int db = 0; db < A_NDB; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dbg[((T == 15) * A_N + wave * 32 + l31) * A_D + db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = o[db][r] * inv;
                }
            }
#endif
            if constexpr (T == 15) if (!T160_ABL(64)) {
                u32x4 xpk[2 * A_NDB];
                pack_o(xpk);
                // the park loads are older than everything hand-over 15 waited for: this wait only tells hipcc where their registers become
                // valid (ONE statement for both wave classes: with it in two branches hipcc resolved the register assignment of one branch
                // by copies placed in FRONT of its wait, i.e. of data that had not landed)
                asm volatile("s_waitcnt vmcnt(6)" : "+v"(ypk[0]), "+v"(ypk[1]), "+v"(ypk[2]), "+v"(ypk[3]), "+v"(ypk[4]), "+v"(ypk[5]), "+v"(ypk[6]), "+v"(ypk[7]), "+v"(ypk[8]), "+v"(ypk[9]) :: "memory");
                float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
                if (!mse) {
#pragma unroll
                    for (int i = 0; i < 2 * A_NDB; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned ux = xpk[i][e], uy = ypk[i][e];      // (bit_cast of a vector-element lvalue reads element 0: go through a scalar)
                            const h16x2 x = __builtin_bit_cast(h16x2, ux), y = __builtin_bit_cast(h16x2, uy);
                            a0[e] = H16_DOT2(x, y, a0[e]);
                            a1[e] = H16_DOT2(x, x, a1[e]);
                            a2[e] = H16_DOT2(y, y, a2[e]);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 2 * A_NDB; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned ux = xpk[i][e], uy = ypk[i][e];      // (bit_cast of a vector-element lvalue reads element 0: go through a scalar)
                            const h16x2 x = __builtin_bit_cast(h16x2, ux), y = __builtin_bit_cast(h16x2, uy);
                            const float d0 = (float)x[0] - (float)y[0], d1 = (float)x[1] - (float)y[1];
                            a0[e] = fmaf(d0, d0, a0[e]);
                            a0[e] = fmaf(d1, d1, a0[e]);
                        }
                }
                float s0 = (a0[0] + a0[1]) + (a0[2] + a0[3]);
                float s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
                float s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
                s0 = wave_sum(s0);
                s1 = wave_sum(s1);
                s2 = wave_sum(s2);
                if (lane == 0) {
                    // (x = the cross attention's output, y = the self attention's: direction 1 ran them in the other order)
                    f32x4 r4 = {s0, dir ? s2 : s1, dir ? s1 : s2, 0.f};
                    *reinterpret_cast<f32x4*>(part + (size_t)cur.pidx * 4) = r4;
                }
            }
        };
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{});
        step(IC<8>{}); step(IC<9>{}); step(IC<10>{}); step(IC<11>{}); step(IC<12>{}); step(IC<13>{}); step(IC<14>{}); step(IC<15>{});
#if defined(DSIM_DEVTOOLS) && defined(DSIM_STAMPS)
        TSTAMP(8);
        tsa_[7] += tst_[8] - tst_[7];               // the epilogues (pack, park, partial sums)
#endif
        if (!has_next) break;
        cur = nxt;
        read_q();           // (its ten pieces were issued in steps 0-9: older than everything step 15's wait left in flight)
        ++it;
    }
    wait_vm<0>();           // the look-ahead pieces of the last unit land in this workgroup's LDS
#if defined(DSIM_DEVTOOLS) && defined(DSIM_STAMPS)
    if (dbg && lane == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(dbg)[((size_t)blockIdx.x * 8 + wave) * 8 + i] = tsa_[i];
#endif
}

// One wave per pair: f64 fold of the per-wave partials in a fixed order (lane i adds entries i, i + 64, ...; xor tree), then
// cosine / mse and the mean of the two directions (diffsim.py:187-197; F.cosine_similarity eps = 1e-8)
__global__ __launch_bounds__(64) void pair_finish160_kernel(const float* __restrict__ part, int nblk, int mse, double count,
                                                            float* __restrict__ out, int32_t* __restrict__ status) {
    const int p = blockIdx.x, lane = threadIdx.x;
    double res = 0.0;
    for (int dir = 0; dir < 2; ++dir) {
        double a = 0.0, x2 = 0.0, y2 = 0.0;
        const float* o = part + ((size_t)p * 2 + dir) * nblk * 4;
        for (int i = lane; i < nblk; i += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(o + 4 * i);
            a += v[0]; x2 += v[1]; y2 += v[2];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off);
            x2 += __shfl_xor(x2, off);
            y2 += __shfl_xor(y2, off);
        }
        if (mse) res += a / count;
        else {
            const double nx = sqrt(x2), ny = sqrt(y2);
            res += a / (fmax(nx, 1e-8) * fmax(ny, 1e-8));
        }
    }
    if (lane == 0) {
        const float sc = (float)(res * 0.5);
        out[p] = sc;
        if (status) status[p] = (sc - sc == 0.0f) ? 0 : 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The same core as a plain scaled-dot-product attention: the U-Net's own self-attentions of the 16 x 16 level (attn1 of the 1280-channel
// transformer blocks: /root/reference/diffsim/hacked_attn.py:74-81 with 256 tokens, 8 heads x 160).  A unit is one (batch element,
// head); its 8 steps are the tail's pass 0.  Differences from the tail: q, k, v and the output carry their own row strides (the fused
// q|k|v projection writes 3 C-wide rows); every wave issues its DMA pieces behind the barrier; the next unit's ten Q pieces ride in
// hand-overs 0-4 (two each), so they are older than what hand-over 7 waits for and the slab can be read right behind step 7; the
// output, normalised and rounded, goes through the wave's Q slab (just emptied into registers) and leaves as 16-byte row segments.
__global__ __launch_bounds__(512, 2) void sdpa160_kernel(const h16* __restrict__ qg, const h16* __restrict__ kg, const h16* __restrict__ vg,
                                                         h16* __restrict__ og, const int ldq, const int ldk, const int ldo, const int B,
                                                         const int H, const float c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int rbq = ldq * 2, rbk = ldk * 2, rbo = ldo * 2;
    const int total = B * H;
    // workgroup -> unit: the eight consecutive units (heads of one batch element when H = 8: neighbours in every token row, they share
    // cache lines) run at the same time on eight workgroups of ONE XCD
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, g8 = (int)gridDim.x >> 6;
    const unsigned recs_q = (unsigned)((A_N - 1) * rbq + A_ROWB), recs_k = (unsigned)((A_N - 1) * rbk + A_ROWB);

    auto piece_voff = [&](int pat, bool swizzle, int ln, int rb) {
        const int f = 64 * pat + ln;
        const int r = (f * 3277) >> 16, pos = f - r * 20;
        const int ch = swizzle ? ((pos & ~3) | ((pos & 3) ^ (r >> 2))) : pos;
        return r * rb + ch * 16;
    };
    int kv_voff[3], kv_grp[3];
    unsigned kv_lds[3];
    bool kv_isv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pj = wave + 8 * j;
        kv_isv[j] = pj >= 10;
        const int p = kv_isv[j] ? pj - 10 : pj;
        const int grp = p >= 5, pat = p - 5 * grp;
        kv_voff[j] = piece_voff(pat, !kv_isv[j], lane, rbk);
        kv_grp[j] = grp;
        kv_lds[j] = (kv_isv[j] ? A_KTILE : 0) + p * 1024;
    }
    const bool three = wave < 4;
    auto issue_kv = [&](unsigned long long uoff, int tile, int slot) {
        const u32x4 dK = make_desc((const char*)kg + uoff, recs_k), dV = make_desc((const char*)vg + uoff, recs_k);
        int rb = rbk;
        unsigned lb = lbase;
        asm volatile("" : "+s"(rb), "+s"(lb));
        const unsigned sb = lb + slot * A_SLOT;
        const int ts = tile * A_KT * rb;
        dma_piece(sb + kv_lds[0], kv_voff[0], dK, ts + kv_grp[0] * 16 * rb);
        dma_piece(sb + kv_lds[1], kv_voff[1], kv_isv[1] ? dV : dK, ts + kv_grp[1] * 16 * rb);
        if (three) dma_piece(sb + kv_lds[2], kv_voff[2], dV, ts + kv_grp[2] * 16 * rb);
    };
    const unsigned qslab = lbase + A_RING + wave * A_KTILE;
    auto issue_q = [&](unsigned long long uoff, int j) {
        const u32x4 dQ = make_desc((const char*)qg + uoff, recs_q);
        int ln = lane, rb = rbq;
        unsigned qs = qslab;
        asm volatile("" : "+v"(ln), "+s"(rb), "+s"(qs));
        dma_piece(qs + j * 1024, piece_voff(j % 5, true, ln, rb), dQ, (wave * 32 + 16 * (j / 5)) * rb);
    };
    const int swz = (l31 >> 2) & 3;
    const int e0 = l31 * A_ROWB + ((half ^ swz) << 4);
    const int e1 = l31 * A_ROWB + (((2 + half) ^ swz) << 4);
    const int vl = A_KTILE + (4 * half + ((lane & 15) >> 2)) * A_ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    struct Unit { unsigned long long oq, ok, oo; };
    auto setup = [&](int u) {
        Unit r;
        const int b = u / H, h = u - b * H;
        r.oq = ((unsigned long long)b * A_N * ldq + h * A_D) * 2ull;
        r.ok = ((unsigned long long)b * A_N * ldk + h * A_D) * 2ull;
        r.oo = ((unsigned long long)(b * A_N + wave * 32) * ldo + h * A_D) * 2ull;
        return r;
    };
    auto unit_of = [&](int it) { return ((it * g8 + (wslot >> 3)) * 8 + xcd) * 8 + (wslot & 7); };
    h16x8 q[A_NKS];
    auto read_q = [&]() {
        const char* s0 = smem + A_RING + wave * A_KTILE;
#pragma unroll
        for (int ks = 0; ks < A_NKS; ++ks) q[ks] = *reinterpret_cast<const h16x8*>(s0 + ((ks & 1) ? e1 : e0) + (ks >> 1) * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    int it = 0;
    if (unit_of(0) >= total) return;
    Unit cur = setup(unit_of(0));
#pragma unroll
    for (int j = 0; j < 10; ++j) issue_q(cur.oq, j);
#pragma unroll
    for (int t = 0; t < A_NSLOT; ++t) issue_kv(cur.ok, t, t);
    if (three) wait_vm<3 * A_NSLOT>(); else wait_vm<2 * A_NSLOT>();
    read_q();
    if (three) wait_vm<3 * (A_NSLOT - 1)>(); else wait_vm<2 * (A_NSLOT - 1)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    h16x8 kpre[A_PRE];
#pragma unroll
    for (int ks = 0; ks < A_PRE; ++ks) kpre[ks] = *reinterpret_cast<const h16x8*>(smem + ((ks & 1) ? e1 : e0) + (ks >> 1) * 64);

    f32x16 o[A_NDB];
    float m_run = 0.f, l_run = 0.f;
    const float thr = A_THR / c;
    bool first = true;                  // the first unit: nothing of a previous unit (its output stores, its late Q pieces) is in flight

    for (;;) {
        const int un = unit_of(it + 1);
        const bool has_next = un < total;
        const Unit nxt = has_next ? setup(un) : cur;       // (no next unit: the look-ahead re-fetches this unit's own rows)

        auto step = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr bool FIRST = T == 0;
            const char* sb = smem + (T & 3) * A_SLOT;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            h16x8 kf[A_PRE + 1];
#pragma unroll
            for (int i = 0; i < A_PRE; ++i) kf[i] = kpre[i];
#pragma unroll
            for (int ks = 0; ks < A_NKS; ++ks) {
                if (ks + A_PRE < A_NKS)
                    kf[(ks + A_PRE) % (A_PRE + 1)] = *reinterpret_cast<const h16x8*>(sb + (((ks + A_PRE) & 1) ? e1 : e0) + ((ks + A_PRE) >> 1) * 64);
                __builtin_amdgcn_sched_barrier(0);
                s = H16_MFMA_32x32x16(kf[ks % (A_PRE + 1)], q[ks], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            const char* vb = sb + vl;
            auto vread = [&](int j) {
                const char* pa = vb + (j / A_NDB) * 16 * A_ROWB + (j % A_NDB) * 64;
                const h16x4 lo = h16_ds_read_tr16_b64(pa);
                const h16x4 hi = h16_ds_read_tr16_b64(pa + 8 * A_ROWB);
                h16x8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                return vf;
            };
            h16x8 vf[A_PRE + 1];
#pragma unroll
            for (int j = 0; j < A_PRE; ++j) vf[j] = vread(j);
            __builtin_amdgcn_sched_barrier(0);
            float tmax = max16_after_mfma(s);
            tmax = max_halves160(tmax);
            if constexpr (FIRST) {
                m_run = tmax;
            } else {
                if (!__all(tmax <= m_run + thr)) {
                    const float mn = fmaxf(m_run, tmax);
                    const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * c);
                    m_run = mn;
                    l_run *= alpha;
#pragma unroll
                    for (int db = 0; db < A_NDB; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                }
            }
            const float mc = -m_run * c;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, mc));
            const float psum = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) +
                               (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
            l_run = FIRST ? psum : l_run + psum;
            h16x8 pf[2];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[f][e] = (h16)s[8 * f + e];
            constexpr int NPV = 2 * A_NDB;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NPV; ++j) {
                if (j + A_PRE < NPV) vf[(j + A_PRE) % (A_PRE + 1)] = vread(j + A_PRE);
                if (j == NPV - A_PRE) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // Issued since the pieces of tile T + 1 (hand-over T - 3): two hand-overs' K / V pieces (3 or 2 each) and Q pieces
                    // (two per hand-over 0-4), and -- in steps 0-2 -- the previous unit's ten output stores.  In the first unit
                    // neither the stores nor the Q pieces of hand-overs 6, 7 exist.
                    constexpr int NQ2 = (((T + 6) & 7) <= 4 ? 2 : 0) + (((T + 7) & 7) <= 4 ? 2 : 0);
                    constexpr int STEADY = NQ2 + (T <= 2 ? 10 : 0);
                    constexpr int FIRSTU = T == 0 ? 0 : T == 1 ? 2 : T == 2 ? 4 : STEADY;
                    if (T <= 2 && first) { if (three) wait_vm<6 + FIRSTU>(); else wait_vm<4 + FIRSTU>(); }
                    else { if (three) wait_vm<6 + STEADY>(); else wait_vm<4 + STEADY>(); }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if constexpr (T <= 4) { issue_q(nxt.oq, 2 * T); issue_q(nxt.oq, 2 * T + 1); }
                    constexpr int TN = T + A_NSLOT;
                    if constexpr (TN < 8) issue_kv(cur.ok, TN, TN & 3);
                    else issue_kv(nxt.ok, TN - 8, TN & 3);
                    const char* sn = smem + ((T + 1) & 3) * A_SLOT;
#pragma unroll
                    for (int ks = 0; ks < A_PRE; ++ks) kpre[ks] = *reinterpret_cast<const h16x8*>(sn + ((ks & 1) ? e1 : e0) + (ks >> 1) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int s2 = j / A_NDB, db = j % A_NDB;
                if (FIRST && s2 == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    o[db] = H16_MFMA_32x32x16(vf[j % (A_PRE + 1)], pf[s2], z, 0, 0, 0);
                } else {
                    o[db] = H16_MFMA_32x32x16(vf[j % (A_PRE + 1)], pf[s2], o[db], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{});
        first = false;

        // ---- the unit's output.  The next unit's Q rows are complete in the slab (their pieces are older than what hand-over 7 waited
        // for): into registers first, then the slab turns the accumulators' [d][query] lanes into rows -- 8-byte chunks in, 16-byte
        // chunks out, chunk index XOR-swizzled by (row >> 1) & 3 inside groups of four so that the 16 rows a store instruction's lane
        // group covers do not meet on two banks
        read_q();
        {
            const float inv = __builtin_amdgcn_rcpf(half_sum(l_run));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            char* slab = smem + A_RING + wave * A_KTILE;
            const int wrow = l31 * A_ROWB + 8 * half, wsw = (l31 >> 1) & 3;
#pragma unroll
            for (int db = 0; db < A_NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const h16x2 v0 = __builtin_convertvector((f32x2){o[db][4 * g] * inv, o[db][4 * g + 1] * inv}, h16x2);
                    const h16x2 v1 = __builtin_convertvector((f32x2){o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv}, h16x2);
                    const u32x2 w = {__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)};
                    *reinterpret_cast<u32x2*>(slab + wrow + ((4 * db + (g ^ wsw)) << 4)) = w;
                }
            const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)og + cur.oo), 0, 31 * rbo + A_ROWB, 0x00020000);
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int f = 64 * (i % 5) + ln;
                const int r5 = (f * 3277) >> 16, pos = f - r5 * 20, r = r5 + 16 * (i / 5);
                const int ch = (pos & ~3) | ((pos & 3) ^ ((r >> 1) & 3));
                const u32x4 d = *reinterpret_cast<const u32x4*>(slab + r * A_ROWB + (ch << 4));
                __builtin_amdgcn_raw_buffer_store_b128(d, rO, r * rbo + (pos << 4), 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slab has been read: the next hand-over's Q pieces may land
        }
        if (!has_next) break;
        cur = nxt;
        ++it;
    }
    wait_vm<0>();
}

// workgroups the launch uses: one per CU, a multiple of 16 (two directions x eight XCDs), no more than the units there are
int tail160_grid(int n_pairs, int B, int H) {
    int g = cu_count() & ~15;
    if (g < 16) g = 16;
    if (g > 512) g = 512;
    const long need = (((long)n_pairs * B * H + 7) / 8) * 16;      // couples per XCD x 16
    if (need < g) g = (int)need;
    return g;
}

}  // namespace

#ifdef DSIM_DEVTOOLS
float* g_tail160_dbg = nullptr;
int g_tail160_exp = 0;
#endif

bool pair_score160_applies(int N, int D, int dtype) { return N == A_N && D == A_D && dtype == DSIM_H16; }

// the U-Net's 256-token, d = 160 self-attention on the persistent core: 16-bit types, one K / V per query batch element, 16-byte rows
bool sdpa160_applies(const AttnArgs& a) {
    if (a.D != A_D || a.Nq != A_N || a.Nk != A_N || a.Bkv != a.B || a.B < 1 || a.H < 1) return false;
    if (a.ldq % 8 || a.ldk % 8 || a.ldo % 8) return false;
    if (((size_t)a.q | (size_t)a.k | (size_t)a.v | (size_t)a.out) & 15) return false;
    // 32-bit buffer offsets inside one (batch element, head) view
    return (long)(A_N - 1) * a.ldq * 2 + A_ROWB < (1l << 31) && (long)(A_N - 1) * a.ldk * 2 + A_ROWB < (1l << 31);
}

int launch_sdpa160(const AttnArgs& a, hipStream_t s) {
    int g = cu_count() & ~63;
    if (g < 64) g = 64;
    if (g > 512) g = 512;
    const long need = (((long)a.B * a.H + 63) / 64) * 64;
    if (need < g) g = (int)need;
    static DeviceOnce once;
    auto kern = sdpa160_kernel;
    CK_ONCE(once, kern, A_LDS);
    const float c = (1.0f / sqrtf((float)A_D)) * 1.4426950408889634f;
    hipLaunchKernelGGL(kern, dim3(g), dim3(512), A_LDS, s, (const h16*)a.q, (const h16*)a.k, (const h16*)a.v, (h16*)a.out, a.ldq, a.ldk,
                       a.ldo, a.B, a.H, c);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

static size_t tail160_part_bytes(int n_pairs, int B, int H) { return (((size_t)n_pairs * 2 * B * H * 8 * 4 * sizeof(float)) + 255) & ~(size_t)255; }

// partial sums + one 80 KB park slab per workgroup (the grid is bounded by 512 workgroups whatever the device, which keeps this
// function free of device queries)
size_t pair_score160_scratch_bytes(int n_pairs, int B, int H) {
    const long need = (((long)n_pairs * B * H + 7) / 8) * 16;
    return tail160_part_bytes(n_pairs, B, H) + (size_t)(need < 512 ? need : 512) * 8 * A_KTILE;
}

int launch_pair_score160(const void* q, const void* k, const void* v, const int32_t* ia, const int32_t* ib, int n_pairs, int B,
                         int H, int mse, float* out, void* scratch, size_t scratch_bytes, hipStream_t s, int32_t* status) {
    if (scratch_bytes < pair_score160_scratch_bytes(n_pairs, B, H)) return DSIM_ERR_WORKSPACE;
    const int grid = tail160_grid(n_pairs, B, H);
    static DeviceOnce once;
    auto kern = pair_tail160_kernel;
    CK_ONCE(once, kern, A_LDS);
    const float c = (1.0f / sqrtf((float)A_D)) * 1.4426950408889634f;
#ifdef DSIM_DEVTOOLS
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), A_LDS, s, (const h16*)q, (const h16*)k, (const h16*)v, ia, ib, n_pairs, B, H, c,
                       mse, (float*)scratch, (char*)scratch + tail160_part_bytes(n_pairs, B, H), g_tail160_dbg, g_tail160_exp);
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), A_LDS, s, (const h16*)q, (const h16*)k, (const h16*)v, ia, ib, n_pairs, B, H, c,
                       mse, (float*)scratch, (char*)scratch + tail160_part_bytes(n_pairs, B, H));
#endif
    hipLaunchKernelGGL(pair_finish160_kernel, dim3(n_pairs), dim3(64), 0, s, (const float*)scratch, B * H * 8, mse,
                       (double)B * H * A_N * A_D, out, status);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

}  // namespace dsim
