// The 256-token, d = 160 attention core of SD1.5's 16 x 16 level for gfx950 -- the shape BASELINE.json's utilisation target is
// stated on ("up_blocks[0] attention GEMM"):
//
//  * pair_tail160_kernel : the DiffSim score tail at the default tap -- /root/reference/diffsim/diffsim.py:177-197:
//                          O_aa = SDPA(Qa, Ka, Va), O_ab = SDPA(Qa, Kb, Vb) (and the b <-> a mirror), cosine / mse over the
//                          flattened (B, H, N, D) tensors -- for N = 256 tokens, head dim 160, 16-bit compute types.
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
//     K and V as 32-key tiles (10 KB + 10 KB) through a 4-slot ring, THREE tiles ahead, one s_barrier per tile behind a
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
constexpr int A_LOOK = A_NSLOT - 1;             // tiles in flight ahead of the one being read
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

template <int N> struct IC { static constexpr int value = N; };

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

struct Unit160 {
    const h16* q;
    const h16* ks; const h16* vs;       // the query image's own keys / values ("self")
    const h16* kx; const h16* vx;       // the other image's ("cross")
    int pidx;                           // this wave's slot in the partial array
};

__global__ __launch_bounds__(512, 2) void pair_tail160_kernel(const h16* __restrict__ qg, const h16* __restrict__ kg,
                                                              const h16* __restrict__ vg, const int32_t* __restrict__ idx_a,
                                                              const int32_t* __restrict__ idx_b, const int n_pairs, const int B,
                                                              const int H, const float c, const int mse, float* __restrict__ part
#ifdef DSIM_DEVTOOLS
                                                              , float* __restrict__ dbg
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
    auto piece_voff = [&](int pat, bool swizzle) {
        const int f = 64 * pat + lane;
        const int r = (f * 3277) >> 16, pos = f - r * 20;            // f / 20, f % 20 (f < 320)
        const int ch = swizzle ? ((pos & ~3) | ((pos & 3) ^ (r >> 2))) : pos;
        return r * rowb + ch * 16;
    };
    // K / V: this wave's pieces of a tile are j = wave, wave + 8, wave + 16 (< 20) of [K pieces 0..9 | V pieces 0..9]: waves 0-3 issue
    // three, waves 4-7 two (the two waves of a SIMD five together)
    int kv_voff[3], kv_soff[3];
    unsigned kv_lds[3];
    bool kv_isv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pj = wave + 8 * j;
        kv_isv[j] = pj >= 10;
        const int p = kv_isv[j] ? pj - 10 : pj;
        const int grp = p >= 5, pat = p - 5 * grp;
        kv_voff[j] = piece_voff(pat, !kv_isv[j]);
        kv_soff[j] = grp * 16 * rowb;
        kv_lds[j] = (kv_isv[j] ? A_KTILE : 0) + p * 1024;
    }
    const bool three = wave < 4;
    auto issue_kv = [&](const h16* kp, const h16* vp, int tile, int slot) {
        const u32x4 dK = make_desc(kp, recs), dV = make_desc(vp, recs);
        const unsigned sb = lbase + slot * A_SLOT;
        const int ts = tile * A_KT * rowb;
        dma_piece(sb + kv_lds[0], kv_voff[0], dK, ts + kv_soff[0]);                                 // j = 0: always a K piece
        dma_piece(sb + kv_lds[1], kv_voff[1], kv_isv[1] ? dV : dK, ts + kv_soff[1]);
        if (three) dma_piece(sb + kv_lds[2], kv_voff[2], dV, ts + kv_soff[2]);                      // j = 2: always a V piece
    };
    // Q: piece j (0..9) of this wave's 32 rows into its slab
    const unsigned qslab = lbase + A_RING + wave * A_KTILE;
    auto issue_q = [&](const h16* qp, int j) {
        const u32x4 dQ = make_desc(qp, recs);
        dma_piece(qslab + j * 1024, piece_voff(j % 5, true), dQ, (wave * 32 + 16 * (j / 5)) * rowb);
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
        const int iq = dir ? ib : ia, ix = dir ? ia : ib;
        const size_t off = (size_t)b * A_N * ld + h * A_D;
        u.q = qg + iq * img + off;
        u.ks = kg + iq * img + off; u.vs = vg + iq * img + off;
        u.kx = kg + ix * img + off; u.vx = vg + ix * img + off;
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
    for (int j = 0; j < 10; ++j) issue_q(cur.q, j);
#pragma unroll
    for (int t = 0; t < A_LOOK; ++t) issue_kv(cur.ks, cur.vs, t, t);
    if (three) wait_vm<3 * A_LOOK>(); else wait_vm<2 * A_LOOK>();
    read_q();

    f32x16 o[A_NDB];
    float m_run = 0.f, l_run = 0.f;
    u32x4 ypk[2 * A_NDB];

    // the pass's output, normalised and rounded to the compute dtype, two values per register (d = 32 db + (r & 3) + 8 (r >> 2) + 4 half)
    auto pack_o = [&](u32x4 (&pk)[2 * A_NDB]) {
        const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int db = 0; db < A_NDB; ++db)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const h16x2 v = __builtin_convertvector((f32x2){o[db][2 * i] * inv, o[db][2 * i + 1] * inv}, h16x2);
                pk[2 * db + (i >> 2)][i & 3] = __builtin_bit_cast(unsigned, v);
            }
    };

    for (;;) {
        const int p2n = ((it + 1) * npc + cj) * 8 + xcd;
        const bool has_next = p2n < total;
        // (no next unit: the look-ahead re-fetches this unit's own tiles and rows -- same instruction counts, no special cases)
        const Unit160 nxt = has_next ? setup(p2n) : cur;

        // One 32-key tile; steps 0-7 are the self pass, 8-15 the cross pass.  Vector-memory operations per step in issue order:
        // in steps 0-9 one Q piece of the next unit, then the K / V pieces of the tile three steps ahead (3 or 2 per wave).
        // Step T reads the tile issued three steps earlier, so its wait leaves everything issued since then in flight.
        auto step = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr int QP = (T >= 2 && T <= 11 ? 1 : 0) + (T >= 1 && T <= 10 ? 1 : 0);     // Q pieces issued in steps T - 2 and T - 1
            if (three) wait_vm<6 + QP>(); else wait_vm<4 + QP>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {
                constexpr int TN = T + A_LOOK;
                if constexpr (T < 10) issue_q(nxt.q, T);
                if constexpr (TN < 16) issue_kv(TN < 8 ? cur.ks : cur.kx, TN < 8 ? cur.vs : cur.vx, TN & 7, TN & 3);
                else issue_kv(nxt.ks, nxt.vs, TN - 16, TN & 3);
            }
            constexpr bool FIRST = (T & 7) == 0;
            const char* sb = smem + (T & 3) * A_SLOT;
            // ---- S^T = K Q^T: keys x this lane's query column, raw (unscaled) logits ----------------------------------------
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < A_NKS; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(sb + ((ks & 1) ? e1 : e0) + (ks >> 1) * 64);
                s = H16_MFMA_32x32x16(kf, q[ks], s, 0, 0, 0);
            }
            // ---- online softmax -------------------------------------------------------------------------------------------
            float tmax = fmaxf(s[0], s[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(s[r], s[r + 1]));
            tmax = max_halves160(tmax);
            if constexpr (FIRST) {
                m_run = tmax;
            } else {
                if (!__all(tmax <= m_run)) {            // some row's maximum grew: exact rescale (alpha == 1 for the other rows)
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
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, mc));
                psum += s[r];
            }
            l_run = FIRST ? psum : l_run + psum;
            h16x8 pf[2];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[f][e] = (h16)s[8 * f + e];
            // ---- O^T += V^T P^T -------------------------------------------------------------------------------------------
            const char* vb = sb + vl;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < A_NDB; ++db) {
                    const char* pa = vb + s2 * 16 * A_ROWB + db * 64;
                    const h16x4 lo = h16_ds_read_tr16_b64(pa);
                    const h16x4 hi = h16_ds_read_tr16_b64(pa + 8 * A_ROWB);
                    h16x8 vf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                    if (FIRST && s2 == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        o[db] = H16_MFMA_32x32x16(vf, pf[s2], z, 0, 0, 0);
                    } else {
                        o[db] = H16_MFMA_32x32x16(vf, pf[s2], o[db], 0, 0, 0);
                    }
                }
            if constexpr (T == 7) pack_o(ypk);
#ifdef DSIM_DEVTOOLS
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
            if constexpr (T == 15) {
                u32x4 xpk[2 * A_NDB];
                pack_o(xpk);
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
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    s0 += __shfl_xor(s0, off);
                    s1 += __shfl_xor(s1, off);
                    s2 += __shfl_xor(s2, off);
                }
                if (lane == 0) {
                    f32x4 r4 = {s0, s1, s2, 0.f};
                    *reinterpret_cast<f32x4*>(part + (size_t)cur.pidx * 4) = r4;
                }
            }
        };
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{});
        step(IC<8>{}); step(IC<9>{}); step(IC<10>{}); step(IC<11>{}); step(IC<12>{}); step(IC<13>{}); step(IC<14>{}); step(IC<15>{});
        if (!has_next) break;
        cur = nxt;
        read_q();           // (its ten pieces were issued in steps 0-9: older than everything step 15's wait left in flight)
        ++it;
    }
    wait_vm<0>();           // the look-ahead pieces of the last unit land in this workgroup's LDS
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
#endif

bool pair_score160_applies(int N, int D, int dtype) { return N == A_N && D == A_D && dtype == DSIM_H16; }

size_t pair_score160_scratch_bytes(int n_pairs, int B, int H) { return (size_t)n_pairs * 2 * B * H * 8 * 4 * sizeof(float); }

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
                       mse, (float*)scratch, g_tail160_dbg);
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), A_LDS, s, (const h16*)q, (const h16*)k, (const h16*)v, ia, ib, n_pairs, B, H, c,
                       mse, (float*)scratch);
#endif
    hipLaunchKernelGGL(pair_finish160_kernel, dim3(n_pairs), dim3(64), 0, s, (const float*)scratch, B * H * 8, mse,
                       (double)B * H * A_N * A_D, out, status);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

}  // namespace dsim
