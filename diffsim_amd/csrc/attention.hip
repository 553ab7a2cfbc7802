// Flash-style attention and the fused DiffSim score tail for gfx950.
//
//  * attn_kernel     : softmax(Q K^T / sqrt(D)) V for the U-Net's self/cross attention layers
//                      (replaces F.scaled_dot_product_attention inside diffusers' AttnProcessor2_0,
//                      restated at /root/reference/diffsim/hacked_attn.py:74-83).
//  * pair_tail_kernel: the DiffSim score tail -- /root/reference/diffsim/diffsim.py:177-197:
//                      O_ab = SDPA(Qa,Kb,Vb), O_aa = SDPA(Qa,Ka,Va) (and the b<->a mirror), then
//                      cosine (or mse) over the flattened (B,H,N,D) tensors.  Both attentions of a
//                      direction share Q and run in one workgroup; O never leaves registers, only
//                      three f32 partial sums per workgroup reach HBM and a second fixed-order pass
//                      folds them (no float atomics => bit-reproducible scores).
//
// Tiling: a workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows and
// sweeps the keys in 64-row tiles shared through LDS.
//   - S^T = K Q^T is computed with K as the MFMA A operand and Q as B ("swapped QK^T"), so a lane
//     holds one query column of S^T in its accumulator registers: the row max is a per-lane
//     reduction over registers plus one exchange between the two lane halves, and the accumulator
//     is directly the B operand of O^T = V^T P^T (no LDS round trip for P).
//   - Q is pre-scaled by log2(e)/sqrt(D) and the S^T accumulators START at -m (the running row
//     max), so P = exp2(acc) needs no subtract and no multiply: per score element the VALU does
//     one v_exp, half a v_max3 and half a v_cvt_pk.  O is rescaled only in tiles where some row's
//     max grew (exact: alpha == 1 for the other rows).
//   - The softmax denominator comes out of the PV MFMAs: when the head dim leaves a spare column
//     in the 32-wide d block (D = 40, 72, 80, 16) the staged V tile carries a column of ones, so
//     row D of O^T accumulates sum(P) and is rescaled together with O.
//   - V^T fragments come from the row-major V tile by ds_read_b64_tr_b16 (h16) / ds_read_b32 (f32).
//   - h16: next tile's global loads are in flight during the current tile's compute (registers
//     -> double-buffered LDS, one barrier per tile).  f32 parity mode: simple single buffer.
// h16 path: v_mfma_f32_32x32x16_bf16; fp32 parity path: v_mfma_f32_32x32x2_f32 (exact f32).
#include "common.h"

namespace dsim {
namespace {

// max over the two 32-lane halves of a wave in every lane: one v_permlane32_swap (gfx950) instead of a ds_bpermute
// round trip through the LDS pipe -- the softmax branches on this value once per key tile
__device__ __forceinline__ float max_halves(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = lower half, r[1] = upper half, in both
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

constexpr int KT = 64;   // kv rows per LDS tile

template <typename T, int D> struct ACfg {
    static constexpr int ES = sizeof(T);
    static constexpr int VEC = 16 / ES;
    static constexpr int NKS = (D + 15) / 16;     // 16-deep k steps over d (QK^T)
    static constexpr int NDB = (D + 31) / 32;     // 32-wide output blocks over d (PV)
    static constexpr int DPL = NDB * 32;          // LDS columns (zero padded)
    static constexpr bool ONES = D < DPL;         // spare column -> ones column gives the row sum
    // K tile: only the NKS*16 columns QK^T reads (h16); row stride an odd number of 16-B slots (ds_read_b128)
    static constexpr int DPLK = (ES == 2) ? NKS * 16 : DPL;
    static constexpr int RS = DPLK * ES + 16;
    // h16, D = 8 (mod 16): K carries a ones column at d = D and Q carries -m there, so S^T comes out of the MFMAs
    // already relative to the running max and the accumulators start at the constant 0 (no per-tile register fill).
    // Any per-row reference cancels in the softmax, so -m rounded to h16 is exact as long as m itself is kept rounded.
    static constexpr bool KONE = (ES == 2) && (D % 16 == 8);
    // V tile row stride.  h16: the transposed reads (ds_read_b64_tr_b16) take, per 32-lane half, a
    // 4-row x 32-column block = 4 rows x 16 dwords; they are conflict-free when the row stride is
    // 16 or 48 dwords mod 64 (four rows tile the 64 banks).  f32: plain ds_read_b32, same as K.
    static constexpr int RSV = (ES == 2) ? (((DPL * 2) % 256 == 64 || (DPL * 2) % 256 == 192) ? DPL * 2 : DPL * 2 + 64) : RS;
    static constexpr int CPR = DPL / VEC;         // 16-B chunks per row
    static constexpr int TILEK = KT * RS;
    static constexpr int TILE = (KT * RS + KT * RSV + 1) / 2;   // average, so that 2*TILE = K tile + V tile
    // double-buffered staging, except for the widest heads: there two tile pairs (83 KB) would leave one workgroup per
    // CU; a single pair lets a second workgroup hide this one's load latency instead
    static constexpr bool PIPE = sizeof(T) == 2 && DPL < 160;
    static constexpr int LDS = (PIPE ? 4 : 2) * TILE;
    // waves per SIMD the register budget is held to (occupancy hides the serial MFMA/VALU phases)
    // (4 workgroups per CU need <= 40 KB of LDS each: true for d <= 48 now that the K tile is 48 columns wide)
    static constexpr int WPS = (sizeof(T) == 2 && DPL <= 64) ? (LDS <= 40 * 1024 ? 4 : 3) : ((sizeof(T) == 2 && DPL <= 96) ? 2 : (sizeof(T) == 2 ? 2 : 1));
};

struct FragF32 { f32x4 lo, hi; };
template <typename T> struct FragOf { typedef h16x8 type; };
template <> struct FragOf<float> { typedef FragF32 type; };

__device__ __forceinline__ void zero_frag(h16x8& f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (h16)0.0f;
}
__device__ __forceinline__ void zero_frag(FragF32& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.lo[i] = f.hi[i] = 0.f;
}
// load 8 consecutive elements and pre-scale them (Q only)
__device__ __forceinline__ void gload_frag_scaled(h16x8& f, const h16* p, float sc) {
    const h16x8 t = *reinterpret_cast<const h16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (h16)((float)t[i] * sc);
}
__device__ __forceinline__ void gload_frag_scaled(FragF32& f, const float* p, float sc) {
    f.lo = *reinterpret_cast<const f32x4*>(p) * sc;
    f.hi = *reinterpret_cast<const f32x4*>(p + 4) * sc;
}
__device__ __forceinline__ void lload_frag(h16x8& f, const char* p) { f = *reinterpret_cast<const h16x8*>(p); }
__device__ __forceinline__ void lload_frag(FragF32& f, const char* p) {
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 16);
}
__device__ __forceinline__ void mma(const h16x8& a, const h16x8& b, f32x16& c) {
    c = H16_MFMA_32x32x16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const FragF32& a, const FragF32& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], c, 0, 0, 0);
}

// Q fragments of this wave's 32 query rows (pre-scaled by log2(e)/sqrt(D)), resident in registers.
template <typename T, int D> struct QFrags { typename FragOf<T>::type f[ACfg<T, D>::NKS]; };
template <typename T, int D> struct OAcc { f32x16 b[ACfg<T, D>::NDB]; };

template <typename T, int D>
__device__ __forceinline__ void load_q(QFrags<T, D>& qf, const T* qrow /*row base + h*D*/, int half, float scale_log2) {
#pragma unroll
    for (int ks = 0; ks < ACfg<T, D>::NKS; ++ks) {
        const int d0 = 16 * ks + 8 * half;
        if (d0 < D) gload_frag_scaled(qf.f[ks], qrow + d0, scale_log2);
        else zero_frag(qf.f[ks]);
    }
}

// One wave's 32 output rows from its O^T accumulators.  A lane owns one query row and, per 32-wide block db, the 8-byte chunks
// d = 32 db + 8 g + 4 half + (0..3): stored as they lie, a wave instruction writes 16 bytes into each of 32 rows -- 32 cache-line
// operations for 512 bytes, and the texture path, not HBM, sets the time (a 77-key cross-attention spent 0.29 of its 0.49 ms on
// them: profiles/r06_experiments.txt item 4).  The two halves of the wave therefore exchange every other chunk first (one
// v_permlane32_swap per register; both lanes of a pair belong to the same row), each lane stores 16 contiguous bytes, and the
// row gets 32 bytes per instruction from half as many instructions.  val(db, r) = the value to store for accumulator register r.
template <int D, typename F>
__device__ __forceinline__ void store_o_rows(h16* orow, int half, bool wide, F&& val) {
    constexpr int NDB = (D + 31) / 32;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    if (wide) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int d0 = db * 32 + 16 * gp;
                if (d0 < D) {
                    unsigned a[2], b[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const h16x2 va = __builtin_convertvector((f32x2){val(db, 8 * gp + 2 * e), val(db, 8 * gp + 2 * e + 1)}, h16x2);
                        const h16x2 vb = __builtin_convertvector((f32x2){val(db, 8 * gp + 4 + 2 * e), val(db, 8 * gp + 4 + 2 * e + 1)}, h16x2);
                        a[e] = __builtin_bit_cast(unsigned, va);
                        b[e] = __builtin_bit_cast(unsigned, vb);
                    }
                    // lower lanes keep chunk 2 gp and receive the partner's; upper lanes keep chunk 2 gp + 1 and receive the partner's
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                    const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
                    const int d = d0 + 8 * half;
                    if (d < D) *reinterpret_cast<u32x4*>(orow + d) = w;
                }
            }
    } else {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * half;
                if (d < D) {
                    h16x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = (h16)val(db, 4 * g + j);
                    *reinterpret_cast<h16x4*>(orow + d) = v4;
                }
            }
    }
}
// 16-byte row segments need 16-byte aligned rows
__device__ __forceinline__ bool wide_rows(const AttnArgs& p) { return p.ldo % 8 == 0 && ((size_t)p.out & 15) == 0; }

// Staging of one KT-row tile of K and V, split in a load half and a store half so the global loads can be
// issued a whole tile ahead of the LDS writes.  Only the D real columns move per tile: the zero padding up
// to DPL columns (and, with ONES, the 1.0 in column D of V) is written ONCE per attend() by tile_init.
// Rows >= Nk of a ragged tile are never stored: they keep zeros or stale finite values, and their scores
// are masked to -inf, so they contribute exactly 0.
template <typename T, int D> struct StageRegs {
    static constexpr int CPRD = D / ACfg<T, D>::VEC;                 // real 16-B chunks per row
    static constexpr int N = (KT * CPRD + 255) / 256;
    u32x4 k[N], v[N];
    unsigned goff[N];       // element offset of this thread's chunk inside a tile (row * ldk + col)
    unsigned loff[N];       // byte offset inside the K tile image; the V image uses lvoff
    unsigned lvoff[N];
    int row[N];             // tile row, or KT when this thread has no chunk in round i
};

template <typename T> __device__ __forceinline__ u32x4 one_chunk();
template <> __device__ __forceinline__ u32x4 one_chunk<h16>() { u32x4 r = {DSIM_H16_ONE_BITS, 0u, 0u, 0u}; return r; }   // 1.0 in element 0
template <> __device__ __forceinline__ u32x4 one_chunk<float>() { u32x4 r = {0x3F800000u, 0u, 0u, 0u}; return r; }

template <typename T, int D>
__device__ __forceinline__ void tile_init(StageRegs<T, D>& sr, char* lds, int ldk, int Nk, int tid) {
    typedef ACfg<T, D> C;
    typedef StageRegs<T, D> SR;
    const u32x4 z = {0u, 0u, 0u, 0u};
    // zero fill is needed for the padding columns and for the never-stored rows of a ragged last tile
    if (D < C::DPL || (Nk % KT) != 0)
        for (int o = tid * 16; o < C::LDS; o += 256 * 16) *reinterpret_cast<u32x4*>(lds + o) = z;
#pragma unroll
    for (int i = 0; i < SR::N; ++i) {
        const int idx = tid + i * 256;
        const int r = idx / SR::CPRD, c = idx - r * SR::CPRD;
        sr.row[i] = idx < KT * SR::CPRD ? r : KT;
        sr.goff[i] = (unsigned)r * (unsigned)ldk + (unsigned)c * C::VEC;
        sr.loff[i] = (unsigned)(r * C::RS + c * 16);
        sr.lvoff[i] = (unsigned)(C::TILEK + r * C::RSV + c * 16);
    }
    if constexpr (C::ONES) {
        __syncthreads();
        constexpr int NBUF = C::PIPE ? 2 : 1;
        for (int i = tid; i < KT * NBUF; i += 256) {
            const int buf = i / KT, r = i - buf * KT;
            *reinterpret_cast<u32x4*>(lds + buf * 2 * C::TILE + C::TILEK + r * C::RSV + (D / C::VEC) * 16) = one_chunk<T>();
            if constexpr (C::KONE)
                *reinterpret_cast<u32x4*>(lds + buf * 2 * C::TILE + r * C::RS + (D / C::VEC) * 16) = one_chunk<T>();
        }
    }
}

template <typename T, int D>
__device__ __forceinline__ void tile_load(StageRegs<T, D>& sr, const T* kb, const T* vb, int ldk, int kv0, int Nk) {
    typedef StageRegs<T, D> SR;
    const T* kt = kb + (size_t)kv0 * ldk;
    const T* vt = vb + (size_t)kv0 * ldk;
#pragma unroll
    for (int i = 0; i < SR::N; ++i) {
        if (kv0 + sr.row[i] < Nk && sr.row[i] < KT) {
            sr.k[i] = *reinterpret_cast<const u32x4*>(kt + sr.goff[i]);
            sr.v[i] = *reinterpret_cast<const u32x4*>(vt + sr.goff[i]);
        }
    }
}
template <typename T, int D>
__device__ __forceinline__ void tile_store(char* lds, const StageRegs<T, D>& sr, int kv0, int Nk) {
    typedef StageRegs<T, D> SR;
#pragma unroll
    for (int i = 0; i < SR::N; ++i) {
        if (kv0 + sr.row[i] < Nk && sr.row[i] < KT) {
            *reinterpret_cast<u32x4*>(lds + sr.loff[i]) = sr.k[i];
            *reinterpret_cast<u32x4*>(lds + sr.lvoff[i]) = sr.v[i];
        }
    }
}

// One full attention of this wave's 32 query rows against Nk keys.  On return o[db][r] holds the
// NORMALISED output O^T[d = db*32 + (r&3)+8(r>>2)+4*half][q = lane&31].  All 256 threads of the
// workgroup must call it together (it contains workgroup barriers).
// FAST: the running maximum is fixed after key tile 0 -- the later tiles compute P = exp2(S - m) without looking at their
// scores at all (no row maximum, no re-base test, no rescale: a third of the loop's non-exp vector instructions).  Softmax is
// invariant to the reference point, so a row whose true maximum lies above m just carries P > 1 and larger sums (f32 / h16
// have the exponent range for it).  Only if the excess passes ~100 (log2 units) can exp2 overflow; the caller detects that from a
// non-finite or absurd denominator and re-runs the block with FAST = false (attend_checked).
template <typename T, int D, bool FAST = false>
__device__ __forceinline__ void attend(const QFrags<T, D>& qfr, const T* kb, const T* vb, int ldk, int Nk, char* lds,
                                       OAcc<T, D>& oacc, float* l_out = nullptr) {
    typedef ACfg<T, D> C;
    typedef typename FragOf<T>::type Frag;
    QFrags<T, D> qloc = qfr;    // (KONE writes -m into the spare d = D slot of its own copy)
    auto& qf = qloc.f;
    auto& o = oacc.b;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = 0.f;          // running row max (log2 units); meaningful after tile 0 (KONE: a h16 value)
    f32x16 minit;               // the S^T accumulators' start value: -m_run (KONE: 0, the maximum rides in Q's spare slot)
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = 0.f;
    float l_run = 0.f;          // used only when !ONES

    const int ntiles = (Nk + KT - 1) / KT;
    StageRegs<T, D> sr;
    __syncthreads();            // a previous attend() of this workgroup may still be reading the buffers
    tile_init<T, D>(sr, lds, ldk, Nk, tid);
    if constexpr (C::PIPE) tile_load<T, D>(sr, kb, vb, ldk, 0, Nk);
    __syncthreads();
    char* const lds0 = lds;
    for (int kt = 0; kt < ntiles; ++kt) {
        if constexpr (C::PIPE) {
            // buffer (kt&1) was last read in iteration kt-2; every wave has passed barrier kt-1 since
            lds = lds0 + (kt & 1) * 2 * C::TILE;
            tile_store<T, D>(lds, sr, kt * KT, Nk);
            __syncthreads();
            if (kt + 1 < ntiles) tile_load<T, D>(sr, kb, vb, ldk, (kt + 1) * KT, Nk);
        } else {
            __syncthreads();                               // previous tile fully consumed
            tile_load<T, D>(sr, kb, vb, ldk, kt * KT, Nk);
            tile_store<T, D>(lds, sr, kt * KT, Nk);
            __syncthreads();
        }

        // ---- S'^T = K Q^T - m for the two 32-row kv blocks (accumulators start at -m) ----------
        f32x16 s[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            s[jb] = minit;                  // -m in every register (KONE: the constant 0): the first MFMA reads it as its C operand
            const char* krow = lds + (jb * 32 + l31) * C::RS + half * 8 * C::ES;
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                Frag kf;
                lload_frag(kf, krow + ks * 16 * C::ES);
                mma(kf, qf[ks], s[jb]);
            }
        }
        if (kt * KT + KT > Nk) {                           // ragged last tile only
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kt * KT + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (kv >= Nk) s[jb][r] = -INFINITY;
                }
        }
        // ---- online softmax (per query column == per lane) ---------------------------------
        float tmax = -INFINITY;
        if (!FAST || kt == 0) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[jb][r]);
            tmax = max_halves(tmax);
        }
        // tmax is relative to m_run.  Tile 0 always re-bases; later tiles only when some row's max
        // grew (the running max settles after a few tiles) -- exact, not a threshold.
        // (KONE re-bases only past a slack of 0.5, so that rounding m to h16 cannot leave a row just above 0 and
        // re-trigger on every tile; P <= 1.42 there)
        constexpr float SLACK = C::KONE ? 0.5f : 0.f;
        if (kt == 0 || (!FAST && !__all(tmax <= SLACK))) {
            float delta = kt == 0 ? tmax : fmaxf(tmax, 0.f);
            if constexpr (C::KONE) {
                if constexpr (sizeof(T) == 2) {
                    const float m_new = (float)(h16)(m_run + delta);      // the value Q can carry exactly
                    delta = m_new - m_run;
                    m_run = m_new;
                    if (half == 1) qf[C::NKS - 1][0] = (h16)(-m_new);       // d = D lives in element 0 of the upper half
                }
            } else {
                m_run += delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) minit[r] = -m_run;
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[jb][r] -= delta;
            if (kt != 0) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                l_run *= alpha;
#pragma unroll
                for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            }
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[jb][r] = __builtin_amdgcn_exp2f(s[jb][r]);
        if constexpr (!C::ONES) {
            float psum = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) psum += s[jb][r];
            l_run += psum;
        }

        // ---- O^T += V^T P^T ---------------------------------------------------------------
        const char* vt = lds + C::TILEK;
        if constexpr (sizeof(T) == 2) {
            // transposed read: per 16-lane group a 4x16 block; lane 4q+p supplies row q, cols 4p..4p+3
            const int i16 = lane & 15, g = lane >> 4;
            const int trow = 4 * (g >> 1) + (i16 >> 2);            // 4*half + q'
            const int tcol = 16 * (g & 1) + 4 * (i16 & 3);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    h16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (h16)s[jb][8 * s2 + j];
                    const char* vbase = vt + (jb * 32 + 16 * s2 + trow) * C::RSV + tcol * 2;
#pragma unroll
                    for (int db = 0; db < C::NDB; ++db) {
                        const char* pa = vbase + db * 64;
                        h16x4 lo = h16_ds_read_tr16_b64((pa));
                        h16x4 hi = h16_ds_read_tr16_b64((pa + 8 * C::RSV));
                        h16x8 vf;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                        o[db] = H16_MFMA_32x32x16(vf, pf, o[db], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const char* vrow = vt + row * C::RSV + l31 * 4;
#pragma unroll
                    for (int db = 0; db < C::NDB; ++db) {
                        const float a = *reinterpret_cast<const float*>(vrow + db * 128);
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[jb][r], o[db], 0, 0, 0);
                    }
                }
            }
        }
    }
    float l_tot;
    if constexpr (C::ONES) {
        // row D of O^T = sum(P): block D/32, in-block row D%32 = (r&3)+8(r>>2)+4*half
        constexpr int RB = D / 32, RR = D % 32;
        constexpr int RH = (RR >> 2) & 1, REG = (RR & 3) + 4 * (RR >> 3);
        const float mine = o[RB][REG];
        const float other = __shfl_xor(mine, 32);
        l_tot = (half == RH) ? mine : other;
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
    if (l_out) *l_out = l_tot;
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= inv;
}

// attend() for TWO 32-row query blocks per wave (a workgroup = 256 queries): every K and V^T fragment read from LDS feeds both
// blocks.  attend() reads one kilobyte of fragments per MFMA -- at full MFMA rate 256 B per clock per CU, twice the LDS port -- so
// the one-block form tops out near half the matrix rate whatever else it does; this one halves the reads and the K / V staging per
// query.  Same arithmetic per row as attend() (same tiles, same order): results are bit-identical to it.  16-bit types without the
// K ones-column trick (d = 64: SDXL), double-buffered staging.
template <typename T, int D, bool FAST>
__device__ __forceinline__ void attend2(const QFrags<T, D> (&qfr)[2], const T* kb, const T* vb, int ldk, int Nk, char* lds,
                                        OAcc<T, D> (&oacc)[2], float (&l_out)[2]) {
    typedef ACfg<T, D> C;
    static_assert(sizeof(T) == 2 && C::PIPE && !C::KONE, "attend2: 16-bit, double-buffered, no K ones column");
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
    f32x16 minit[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[q].b[db][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) minit[q][r] = 0.f;
    }
    const int ntiles = (Nk + KT - 1) / KT;
    StageRegs<T, D> sr;
    __syncthreads();
    tile_init<T, D>(sr, lds, ldk, Nk, tid);
    tile_load<T, D>(sr, kb, vb, ldk, 0, Nk);
    __syncthreads();
    char* const lds0 = lds;
    const int i16 = lane & 15, g = lane >> 4;
    const int trow = 4 * (g >> 1) + (i16 >> 2), tcol = 16 * (g & 1) + 4 * (i16 & 3);
    for (int kt = 0; kt < ntiles; ++kt) {
        lds = lds0 + (kt & 1) * 2 * C::TILE;
        tile_store<T, D>(lds, sr, kt * KT, Nk);
        __syncthreads();
        if (kt + 1 < ntiles) tile_load<T, D>(sr, kb, vb, ldk, (kt + 1) * KT, Nk);
        f32x16 s[2][2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            s[0][jb] = minit[0];
            s[1][jb] = minit[1];
            const char* krow = lds + (jb * 32 + l31) * C::RS + half * 8 * C::ES;
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                h16x8 kf;
                lload_frag(kf, krow + ks * 16 * C::ES);
                mma(kf, qfr[0].f[ks], s[0][jb]);
                mma(kf, qfr[1].f[ks], s[1][jb]);
            }
        }
        if (kt * KT + KT > Nk) {                           // ragged last tile only
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kt * KT + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (kv >= Nk) s[q][jb][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float tmax = -INFINITY;
            if (!FAST || kt == 0) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[q][jb][r]);
                tmax = max_halves(tmax);
            }
            if (kt == 0 || (!FAST && !__all(tmax <= 0.f))) {
                const float delta = kt == 0 ? tmax : fmaxf(tmax, 0.f);
                m_run[q] += delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) minit[q][r] = -m_run[q];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[q][jb][r] -= delta;
                if (kt != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    l_run[q] *= alpha;
#pragma unroll
                    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[q].b[db][r] *= alpha;
                }
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[q][jb][r] = __builtin_amdgcn_exp2f(s[q][jb][r]);
            if constexpr (!C::ONES) {
                float psum = 0.f;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) psum += s[q][jb][r];
                l_run[q] += psum;
            }
        }
        const char* vt = lds + C::TILEK;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                h16x8 pf[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[q][j] = (h16)s[q][jb][8 * s2 + j];
                const char* vbase = vt + (jb * 32 + 16 * s2 + trow) * C::RSV + tcol * 2;
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) {
                    const char* pa = vbase + db * 64;
                    const h16x4 lo = h16_ds_read_tr16_b64((pa));
                    const h16x4 hi = h16_ds_read_tr16_b64((pa + 8 * C::RSV));
                    h16x8 vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                    oacc[0].b[db] = H16_MFMA_32x32x16(vf, pf[0], oacc[0].b[db], 0, 0, 0);
                    oacc[1].b[db] = H16_MFMA_32x32x16(vf, pf[1], oacc[1].b[db], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        auto& o = oacc[q].b;
        float l_tot;
        if constexpr (C::ONES) {
            constexpr int RB = D / 32, RR = D % 32;
            constexpr int RH = (RR >> 2) & 1, REG = (RR & 3) + 4 * (RR >> 3);
            const float mine = o[RB][REG];
            const float other = __shfl_xor(mine, 32);
            l_tot = (half == RH) ? mine : other;
        } else {
            l_tot = l_run[q] + __shfl_xor(l_run[q], 32);
        }
        l_out[q] = l_tot;
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= inv;
    }
}

// The fast form, checked: every row's denominator must be finite and sane (it is >= ~1 by construction: the row's own
// tile-0 maximum contributes exp2(0)); otherwise some exp2 overflowed and the whole workgroup repeats the block exactly.
// h16 only (the fp32 parity mode keeps the exact running maximum).
template <typename T, int D, bool FASTK>
__device__ __forceinline__ void attend_checked(const QFrags<T, D>& qfr, const T* kb, const T* vb, int ldk, int Nk, char* lds,
                                               OAcc<T, D>& oacc) {
    if constexpr (sizeof(T) == 2 && FASTK) {
        float l = 0.f;
        attend<T, D, true>(qfr, kb, vb, ldk, Nk, lds, oacc, &l);
        // workgroup-uniform decision (attend() contains workgroup barriers): any wave with a bad row sends everybody back
        const int bad = !(l > 0.25f && l < DSIM_H16_LSUM_MAX);
        if (__syncthreads_or(bad)) attend<T, D, false>(qfr, kb, vb, ldk, Nk, lds, oacc);
    } else {
        attend<T, D, false>(qfr, kb, vb, ldk, Nk, lds, oacc);
    }
}

// ---- long key sequences: two query blocks per wave, software-pipelined ---------------------------------------------------
// The 64 x 64 self-attention (4096 keys, d = 40) keeps the MFMA pipe 55 % and the VALU 57 % busy in attn_kernel, but only
// 22 % of the time both at once (r03 counters): within a wave the tile body is serial -- QK^T MFMAs, then 32 v_exp per lane,
// then PV MFMAs -- and the other waves of the SIMD are as likely to be in the same phase as in the complementary one.
// Here the overlap is built into ONE wave's instruction stream.  A wave owns 64 query rows as two 32-column blocks; a key
// tile is four "units" (query block q, key half j), each QK (NKS MFMAs) -> softmax (16 v_exp + 8 v_cvt_pk per lane) -> PV
// (2 NDB MFMAs).  The units run as a three-stage pipeline, one step per unit:
//     step n:   MFMA pipe: PV(unit n-1), QK(unit n+1)     VALU: softmax(unit n)
// with the v_exp / v_cvt of unit n issued between the MFMAs (pinned by sched_barrier), so the matrix pipe executes while
// the VALU converts.  The pipeline crosses tile borders (the last step of tile t starts tile t + 1's first QK and the first
// step of tile t + 1 finishes tile t's last PV), so three K/V tiles are live: a 3-deep LDS ring (58 KB, two workgroups per
// CU), one barrier per tile.  h16, fixed-reference softmax (attend<.., FAST>: the maximum of key tile 0), Nk % 64 == 0.
// DBG: kbench ablation masks (1 no v_exp, 2 no PV MFMAs, 4 no QK MFMAs, 8 no global loads in the loop, 16 no v_cvt); 0 in the product
template <int D, int DBG = 0>
__device__ __forceinline__ void attend_pipelined2(const QFrags<h16, D> (&qfr)[2], const h16* kb, const h16* vb, int ldk, int Nk,
                                                  char* lds0, OAcc<h16, D> (&oacc)[2], float (&l_out)[2]) {
    typedef h16 T;
    typedef ACfg<T, D> C;
    constexpr int BUF = 2 * C::TILE;                // one K tile + one V tile
    constexpr int NM = 2 * C::NDB + C::NKS;         // MFMAs per step
    QFrags<T, D> qloc[2] = {qfr[0], qfr[1]};
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int koff = l31 * C::RS + half * 16;                                                   // K fragment row of this lane
    const int voff = C::TILEK + (4 * (g4 >> 1) + (i16 >> 2)) * C::RSV + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2;   // transposed V read
    f32x16 minit[2];
    float l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb].b[db][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) minit[qb][r] = 0.f;
    }
    const int ntiles = Nk / KT;
    StageRegs<T, D> sr;
    __syncthreads();            // a previous user of the buffers (the exact fallback never runs before this)
    {
        // ring init: zero padding columns, the ones column of V (row sums out of the PV MFMAs) and of K (KONE)
        typedef StageRegs<T, D> SR;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (D < C::DPL)
            for (int o = tid * 16; o < 3 * BUF; o += 256 * 16) *reinterpret_cast<u32x4*>(lds0 + o) = z;
#pragma unroll
        for (int i = 0; i < SR::N; ++i) {
            const int idx = tid + i * 256;
            const int r = idx / SR::CPRD, c = idx - r * SR::CPRD;
            sr.row[i] = idx < KT * SR::CPRD ? r : KT;
            sr.goff[i] = (unsigned)r * (unsigned)ldk + (unsigned)c * C::VEC;
            sr.loff[i] = (unsigned)(r * C::RS + c * 16);
            sr.lvoff[i] = (unsigned)(C::TILEK + r * C::RSV + c * 16);
        }
        if constexpr (C::ONES) {
            __syncthreads();
            for (int i = tid; i < KT * 3; i += 256) {
                const int buf = i / KT, r = i - buf * KT;
                *reinterpret_cast<u32x4*>(lds0 + buf * BUF + C::TILEK + r * C::RSV + (D / C::VEC) * 16) = one_chunk<T>();
                if constexpr (C::KONE) *reinterpret_cast<u32x4*>(lds0 + buf * BUF + r * C::RS + (D / C::VEC) * 16) = one_chunk<T>();
            }
        }
    }
    // Nk % KT == 0: every tile is whole, so the staging needs no per-row bounds test -- the only predicate left is "this thread
    // has a chunk in round i" (KT * CPRD chunks over 256 threads), which is WAVE-uniform (a scalar branch, no exec masking).
    // Past the end the last tile is refetched into a dead ring slot.
    constexpr int NCH = KT * StageRegs<T, D>::CPRD;
    static_assert(NCH % 64 == 0, "whole waves per staging round");
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto stage_load = [&](int t) {
        const int tt = t < ntiles ? t : ntiles - 1;
        const T* kt = kb + (size_t)tt * KT * ldk;
        const T* vt = vb + (size_t)tt * KT * ldk;
#pragma unroll
        for (int i = 0; i < StageRegs<T, D>::N; ++i)
            if (i * 256 + wv * 64 < NCH) {
                sr.k[i] = *reinterpret_cast<const u32x4*>(kt + sr.goff[i]);
                sr.v[i] = *reinterpret_cast<const u32x4*>(vt + sr.goff[i]);
            }
    };
    auto stage_store = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < StageRegs<T, D>::N; ++i)
            if (i * 256 + wv * 64 < NCH) {
                *reinterpret_cast<u32x4*>(buf + sr.loff[i]) = sr.k[i];
                *reinterpret_cast<u32x4*>(buf + sr.lvoff[i]) = sr.v[i];
            }
    };
    stage_load(0);
    __syncthreads();
    stage_store(lds0);
    stage_load(1);
    __syncthreads();

    auto qk = [&](const char* buf, int q, int j, f32x16& s) {            // S'^T block of unit (q, j)
        s = minit[q];
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) {
            const h16x8 kf = *reinterpret_cast<const h16x8*>(buf + koff + j * 32 * C::RS + ks * 32);
            mma(kf, qloc[q].f[ks], s);
        }
    };
    auto vfrag = [&](const char* buf, int j, int s2, int db) {
        const char* pa = buf + voff + (j * 32 + 16 * s2) * C::RSV + db * 64;
        const h16x4 lo = h16_ds_read_tr16_b64((pa));
        const h16x4 hi = h16_ds_read_tr16_b64((pa + 8 * C::RSV));
        h16x8 vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
        return vf;
    };
    auto pv = [&](const char* buf, int q, int j, const h16x8 (&pf)[2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
                oacc[q].b[db] = H16_MFMA_32x32x16(vfrag(buf, j, s2, db), pf[s2], oacc[q].b[db], 0, 0, 0);
    };
    auto softmax = [&](int q, f32x16& s, h16x8 (&pf)[2]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
        if constexpr (!C::ONES) {
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += s[r];
            l_run[q] += psum;
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[f][e] = (h16)s[8 * f + e];
    };

    // ---- key tile 0, unpipelined: it fixes every row's reference point -------------------------------------------------
    h16x8 pfA[2], pfB[2];
    f32x16 sA, sB;
    {
        f32x16 s0[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) qk(lds0, q, j, s0[q][j]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float tmax = -INFINITY;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s0[q][j][r]);
            tmax = max_halves(tmax);
            float delta = tmax;
            if constexpr (C::KONE) {
                const float m_new = (float)(h16)delta;            // the value Q's spare slot carries exactly
                delta = m_new;
                if (half == 1) qloc[q].f[C::NKS - 1][0] = (h16)(-m_new);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) minit[q][r] = -delta;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s0[q][j][r] -= delta;
        }
        h16x8 pft[2];
        softmax(0, s0[0][0], pft); pv(lds0, 0, 0, pft);
        softmax(0, s0[0][1], pft); pv(lds0, 0, 1, pft);
        softmax(1, s0[1][0], pft); pv(lds0, 1, 0, pft);
        softmax(1, s0[1][1], pfB);                          // its PV is the first step's
    }
    // The four steps of a tile need only two fragment sets: steps 1 and 2 share X = {V^T(t, j0), K(t, j1)}, step 3 and the
    // next tile's step 0 share Y = {V^T(t, j1), K(t + 1, j0)}.  Each is read from LDS ONE step before its first use (X during
    // step 0, Y during step 2), so no MFMA waits on an LDS round trip and every fragment read feeds two query blocks.
    struct FragSet { h16x8 vf[2][C::NDB]; h16x8 kf[C::NKS]; };
    auto load_set = [&](FragSet& F, const char* vbuf, int pj, const char* kbuf, int kj) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < C::NDB; ++db) F.vf[s2][db] = vfrag(vbuf, pj, s2, db);
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) F.kf[ks] = *reinterpret_cast<const h16x8*>(kbuf + koff + kj * 32 * C::RS + ks * 32);
    };
    FragSet X, Y;
    // tile 1 becomes visible, tile 2 is on its way; the pipeline's first QK
    stage_store(lds0 + BUF);
    stage_load(2);
    __syncthreads();
    load_set(Y, lds0, 1, lds0 + BUF, 0);
    sA = minit[0];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) mma(Y.kf[ks], qloc[0].f[ks], sA);

    // One pipeline step: the MFMAs of PV(query block pq; P fragments pin; V^T fragments F.vf) and of QK(query block kq; F.kf)
    // -> sout, with the softmax of sin -> pout (a unit of query block sq) issued between them.
    auto step = [&](const FragSet& F, int pq, const h16x8 (&pin)[2], int kq, f32x16& sout, int sq, f32x16& sin, h16x8 (&pout)[2]) {
        __builtin_amdgcn_sched_barrier(0);
        constexpr int EPG = (DBG & 96) == 32 ? 4 : ((DBG & 96) == 64 ? 6 : ((DBG & 96) == 96 ? 8 : (16 + NM - 2) / (NM - 1)));       // v_exp per MFMA over the first NM - 1 groups
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            // QK first: its result is the NEXT step's softmax input (a whole PV group of slack), and the P fragments this step's
            // PV reads were converted a QK group ago
            if (g >= C::NKS) {
                const int s2 = (g - C::NKS) / C::NDB, db = (g - C::NKS) % C::NDB;
                if (!(DBG & 2)) oacc[pq].b[db] = H16_MFMA_32x32x16(F.vf[s2][db], pin[s2], oacc[pq].b[db], 0, 0, 0);
            } else {
                const int ks = g;
                if (DBG & 4) { if (ks == 0) sout = minit[kq]; }
                else if (ks == 0) sout = H16_MFMA_32x32x16(F.kf[0], qloc[kq].f[0], minit[kq], 0, 0, 0);
                else sout = H16_MFMA_32x32x16(F.kf[ks], qloc[kq].f[ks], sout, 0, 0, 0);
            }
#pragma unroll
            for (int r = g * EPG; r < (g + 1) * EPG && r < 16; ++r)
                if (!(DBG & 1)) sin[r] = __builtin_amdgcn_exp2f(sin[r]);
            // the conversions of a P fragment follow its eight exponentials
#pragma unroll
            for (int f = 0; f < 2; ++f)
                if ((g == (8 * f + 7) / EPG + 1) || (g == NM - 1 && (8 * f + 7) / EPG + 1 > NM - 1)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (!(DBG & 16)) pout[f][e] = (h16)sin[8 * f + e];
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!C::ONES) {
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += sin[r];
            l_run[sq] += psum;
        }
    };

    const char* bprev = lds0;              // tile t - 1
    const char* bcur = lds0 + BUF;         // tile t
    char* bnext = lds0 + 2 * BUF;          // tile t + 1
    for (int t = 1; t < ntiles; ++t) {
        // tile t + 1 into the ring slot tile t - 2 left (its last reader was the Y load of iteration t - 2, two barriers
        // ago); rows past Nk are never stored, and the last iteration's QK of "tile ntiles" reads stale finite data whose
        // result nobody uses
        stage_store(bnext);
        if (!(DBG & 8)) stage_load(t + 2);
        __syncthreads();
        load_set(X, bcur, 0, bcur, 1);
        step(Y, 1, pfB, 1, sB, 0, sA, pfA);       // PV(q1, j1, t - 1)   QK(q1, j0, t)       softmax(q0, j0, t)
        step(X, 0, pfA, 0, sA, 1, sB, pfB);       // PV(q0, j0, t)       QK(q0, j1, t)       softmax(q1, j0, t)
        load_set(Y, bcur, 1, bnext, 0);
        step(X, 1, pfB, 1, sB, 0, sA, pfA);       // PV(q1, j0, t)       QK(q1, j1, t)       softmax(q0, j1, t)
        step(Y, 0, pfA, 0, sA, 1, sB, pfB);       // PV(q0, j1, t)       QK(q0, j0, t + 1)   softmax(q1, j1, t)
        const char* tmp = bprev;
        bprev = bcur; bcur = bnext; bnext = const_cast<char*>(tmp);
    }
    // the last tile's last unit: PV(q1, j1) from the Y fragments
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
            oacc[1].b[db] = H16_MFMA_32x32x16(Y.vf[s2][db], pfB[s2], oacc[1].b[db], 0, 0, 0);
    (void)bprev;

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        auto& o = oacc[qb].b;
        float l_tot;
        if constexpr (C::ONES) {
            constexpr int RB = D / 32, RR = D % 32;
            constexpr int RH = (RR >> 2) & 1, REG = (RR & 3) + 4 * (RR >> 3);
            const float mine = o[RB][REG];
            const float other = __shfl_xor(mine, 32);
            l_tot = (half == RH) ? mine : other;
        } else {
            l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        }
        l_out[qb] = l_tot;
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= inv;
    }
}

// grid ceil(Nq/256) * H * B (1-D): a workgroup = 4 waves x 64 query rows
template <int D, int DBG>
__global__ __launch_bounds__(256, 2) void attn_long_kernel(const AttnArgs p, const float scale_log2) {
    typedef h16 T;
    typedef ACfg<T, D> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int nqb = (p.Nq + 255) / 256;
    int bid = blockIdx.x;
    if (p.xcd_remap) {          // as attn_kernel: the query blocks of one (batch, head) run on one XCD back to back
        const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, r = nwg & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
    }
    const int qblk = bid % nqb, bh = bid / nqb;
    const int h = bh % p.H, b = bh / p.H;
    QFrags<T, D> qf[2];
    int q[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        q[qb] = qblk * 256 + wave * 64 + qb * 32 + l31;
        const int qc = q[qb] < p.Nq ? q[qb] : p.Nq - 1;
        load_q<T, D>(qf[qb], (const T*)p.q + ((size_t)b * p.Nq + qc) * p.ldq + h * D, half, scale_log2);
    }
    const size_t kvoff = (size_t)(b % p.Bkv) * p.Nk * p.ldk + h * D;
    OAcc<T, D> oa[2];
    float l[2];
    attend_pipelined2<D, DBG>(qf, (const T*)p.k + kvoff, (const T*)p.v + kvoff, p.ldk, p.Nk, smem, oa, l);
    // every row's denominator must be finite and sane (attend_checked): else the workgroup repeats the block exactly
    const int bad = !(l[0] > 0.25f && l[0] < DSIM_H16_LSUM_MAX) || !(l[1] > 0.25f && l[1] < DSIM_H16_LSUM_MAX);
    if (__syncthreads_or(bad)) {
        attend<T, D, false>(qf[0], (const T*)p.k + kvoff, (const T*)p.v + kvoff, p.ldk, p.Nk, smem, oa[0]);
        attend<T, D, false>(qf[1], (const T*)p.k + kvoff, (const T*)p.v + kvoff, p.ldk, p.Nk, smem, oa[1]);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (q[qb] >= p.Nq) continue;
        T* orow = (T*)p.out + ((size_t)b * p.Nq + q[qb]) * p.ldo + h * D;
        store_o_rows<D>(orow, half, wide_rows(p), [&](int db, int r) { return oa[qb].b[db][r]; });
    }
}

// grid ceil(Nq/256) * H * B (1-D): attend2 -- a workgroup = 4 waves x 64 query rows (two 32-row blocks per wave)
template <int D, bool FASTK>
__global__ __launch_bounds__(256, 2) void attn_q2_kernel(const AttnArgs p, const float scale_log2) {
    typedef h16 T;
    typedef ACfg<T, D> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int nqb = (p.Nq + 255) / 256;
    int bid = blockIdx.x;
    if (p.xcd_remap) {
        const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, r = nwg & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
    }
    const int qblk = bid % nqb, bh = bid / nqb;
    const int h = bh % p.H, b = bh / p.H;
    QFrags<T, D> qf[2];
    int q[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        q[qb] = qblk * 256 + wave * 64 + qb * 32 + l31;
        const int qc = q[qb] < p.Nq ? q[qb] : p.Nq - 1;
        load_q<T, D>(qf[qb], (const T*)p.q + ((size_t)b * p.Nq + qc) * p.ldq + h * D, half, scale_log2);
    }
    const size_t kvoff = (size_t)(b % p.Bkv) * p.Nk * p.ldk + h * D;
    OAcc<T, D> oa[2];
    float l[2];
    attend2<T, D, FASTK>(qf, (const T*)p.k + kvoff, (const T*)p.v + kvoff, p.ldk, p.Nk, smem, oa, l);
    if constexpr (FASTK) {          // as attend_checked: a non-finite or absurd denominator sends the workgroup through the exact form
        const int bad = !(l[0] > 0.25f && l[0] < DSIM_H16_LSUM_MAX) || !(l[1] > 0.25f && l[1] < DSIM_H16_LSUM_MAX);
        if (__syncthreads_or(bad)) attend2<T, D, false>(qf, (const T*)p.k + kvoff, (const T*)p.v + kvoff, p.ldk, p.Nk, smem, oa, l);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (q[qb] >= p.Nq) continue;
        T* orow = (T*)p.out + ((size_t)b * p.Nq + q[qb]) * p.ldo + h * D;
        store_o_rows<D>(orow, half, wide_rows(p), [&](int db, int r) { return oa[qb].b[db][r]; });
    }
}

// grid ceil(Nq/128) * H * B (1-D)
// FASTK: the fixed-reference softmax of attend<.., FAST> (long key sequences; see launch_attn_d)
template <typename T, int D, bool FASTK>
__global__ __launch_bounds__(256, (ACfg<T, D>::WPS)) void attn_kernel(const AttnArgs p, const float scale_log2) {
    typedef ACfg<T, D> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Give every XCD a
    // contiguous run of logical blocks so that the query blocks of one (batch, head) -- which all re-read the same
    // K and V -- run on ONE XCD back to back and find them in its L2 (bijective for any grid size).
    const int nqb = (p.Nq + 127) / 128;
    int bid = blockIdx.x;
    if (p.xcd_remap) {
        const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, r = nwg & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
    }
    const int qblk = bid % nqb, bh = bid / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int q = qblk * 128 + wave * 32 + l31;
    const int qc = q < p.Nq ? q : p.Nq - 1;
    const T* qrow = (const T*)p.q + ((size_t)b * p.Nq + qc) * p.ldq + h * D;
    QFrags<T, D> qf;
    load_q<T, D>(qf, qrow, half, scale_log2);
    const size_t kvoff = (size_t)(b % p.Bkv) * p.Nk * p.ldk + h * D;
    OAcc<T, D> oa;
    attend_checked<T, D, FASTK>(qf, (const T*)p.k + kvoff, (const T*)p.v + kvoff, p.ldk, p.Nk, smem, oa);
    auto& o = oa.b;
    if (q < p.Nq) {
        T* orow = (T*)p.out + ((size_t)b * p.Nq + q) * p.ldo + h * D;
        if constexpr (sizeof(T) == 2) {
            store_o_rows<D>((h16*)orow, half, wide_rows(p), [&](int db, int r) { return o[db][r]; });
        } else {
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * half;
                    if (d < D) {
                        f32x4 v4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = o[db][4 * g + j];
                        *reinterpret_cast<f32x4*>(orow + d) = v4;
                    }
                }
        }
    }
}

// ---- short key sequences (the 77-key prompt context of every cross-attention) -------------------------------------------
// attn_kernel spends most of such a launch around its two key tiles: 38 KB of LDS zero-fill, the staging of K and V, three
// barriers and an online-softmax loop whose second tile is four fifths padding -- per 128 queries, 65 536 workgroups at the
// 64 x 64 level (181 TF/s, 2.4 TB/s).  Here the keys (<= 96 = three 32-row blocks) are staged ONCE per workgroup and stay in
// LDS while the workgroup walks QIT query blocks of its (batch, head) with no barrier in the loop: per block three S^T tiles,
// ONE softmax pass with the exact row maximum (no running state), PV, normalise, store.  The launch becomes what its bytes say
// it is: a stream of Q in and O out.  h16; K / V images in attn_kernel's padded row-major layouts (ones column of V included).
// The heads of a token share cache lines in Q and O (80 bytes per head at d = 40): the XCD-aware block order that keeps the
// heads of a batch element on one XCD matters more than anything inside the loop (0.62 -> 0.46 ms at d = 40).
#ifdef DSIM_DEVTOOLS
// (kbench only: round 5's form of the kernel below, for interleaved A/B -- g_attn_short = 2)
template <int D>
__global__ __launch_bounds__(256, (D <= 80 ? 4 : 2)) void attn_short_v1_kernel(const AttnArgs p, const float scale_log2, const int qit) {
    typedef h16 T;
    typedef ACfg<T, D> C;
    constexpr int KR = 96;                                   // key rows held (three 32-row MFMA blocks)
    constexpr int CPRD = D / C::VEC;                         // real 16-byte chunks per row
    constexpr int VOFF = KR * C::RS;                         // V image behind the K image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int nqc = (p.Nq + 128 * qit - 1) / (128 * qit);    // query chunks per (batch, head)
    int bid = blockIdx.x;
    if (p.xcd_remap) {          // as attn_kernel: every XCD a contiguous run of (batch, head, chunk) items -- the heads of a row share cache lines
        const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, r = nwg & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
    }
    const int qc = bid % nqc, bh = bid / nqc;
    const int h = bh % p.H, b = bh / p.H;
    const size_t kvoff = (size_t)(b % p.Bkv) * p.Nk * p.ldk + h * D;
    const T* kb = (const T*)p.k + kvoff;
    const T* vb = (const T*)p.v + kvoff;
    // ---- stage K and V once: zero image (padding columns, rows >= Nk), then the real chunks and V's ones column ---------
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int o = tid * 16; o < KR * (C::RS + C::RSV); o += 256 * 16) *reinterpret_cast<u32x4*>(smem + o) = z;
        __syncthreads();
        for (int idx = tid; idx < p.Nk * CPRD; idx += 256) {
            const int r = idx / CPRD, c = idx - r * CPRD;
            const u32x4 kk = *reinterpret_cast<const u32x4*>(kb + (size_t)r * p.ldk + c * C::VEC);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(vb + (size_t)r * p.ldk + c * C::VEC);
            *reinterpret_cast<u32x4*>(smem + r * C::RS + c * 16) = kk;
            *reinterpret_cast<u32x4*>(smem + VOFF + r * C::RSV + c * 16) = vv;
        }
        if constexpr (C::ONES)
            for (int r = tid; r < p.Nk; r += 256) *reinterpret_cast<u32x4*>(smem + VOFF + r * C::RSV + CPRD * 16) = one_chunk<T>();
        __syncthreads();
    }
    const int i16 = lane & 15, g4 = lane >> 4;
    const char* const kfr = smem + l31 * C::RS + half * 16;
    const char* const vfr = smem + VOFF + (4 * (g4 >> 1) + (i16 >> 2)) * C::RSV + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2;
    const int q0 = qc * 128 * qit + wave * 32 + l31;
    QFrags<T, D> qf;
    {
        const int qq = q0 < p.Nq ? q0 : p.Nq - 1;
        load_q<T, D>(qf, (const T*)p.q + ((size_t)b * p.Nq + qq) * p.ldq + h * D, half, scale_log2);
    }
    for (int it = 0; it < qit; ++it) {
        const int q = q0 + it * 128;
        if (q - l31 >= p.Nq) break;                          // wave-uniform: this wave's rows are past the end
        // the next block's Q rows fly while this block computes
        QFrags<T, D> qn;
        {
            const int qq = q + 128 < p.Nq ? q + 128 : p.Nq - 1;
            load_q<T, D>(qn, (const T*)p.q + ((size_t)b * p.Nq + qq) * p.ldq + h * D, half, scale_log2);
        }
        f32x16 s[3];
#pragma unroll
        for (int kbk = 0; kbk < 3; ++kbk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kbk][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                h16x8 kf;
                lload_frag(kf, kfr + kbk * 32 * C::RS + ks * 32);
                mma(kf, qf.f[ks], s[kbk]);
            }
        }
        // keys >= Nk never count; exact row maximum over the (<= 96) keys: lane-local + one exchange between the halves
        float m = -INFINITY;
#pragma unroll
        for (int kbk = 0; kbk < 3; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kv >= p.Nk) s[kbk][r] = -INFINITY;
                m = fmaxf(m, s[kbk][r]);
            }
        m = max_halves(m);
        float psum = 0.f;
#pragma unroll
        for (int kbk = 0; kbk < 3; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kbk][r] = __builtin_amdgcn_exp2f(s[kbk][r] - m);
                if constexpr (!C::ONES) psum += s[kbk][r];
            }
        f32x16 o[C::NDB];
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
        for (int kbk = 0; kbk < 3; ++kbk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                h16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (h16)s[kbk][8 * s2 + e];
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) {
                    const char* pa = vfr + (kbk * 32 + 16 * s2) * C::RSV + db * 64;
                    const h16x4 lo = h16_ds_read_tr16_b64((pa));
                    const h16x4 hi = h16_ds_read_tr16_b64((pa + 8 * C::RSV));
                    h16x8 vf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                    o[db] = H16_MFMA_32x32x16(vf, pf, o[db], 0, 0, 0);
                }
            }
        float l_tot;
        if constexpr (C::ONES) {
            constexpr int RB = D / 32, RR = D % 32;
            constexpr int RH = (RR >> 2) & 1, REG = (RR & 3) + 4 * (RR >> 3);
            const float mine = o[RB][REG];
            const float other = __shfl_xor(mine, 32);
            l_tot = (half == RH) ? mine : other;
        } else {
            l_tot = psum + __shfl_xor(psum, 32);
        }
        const float inv = 1.0f / l_tot;
        if (q < p.Nq) {
            T* orow = (T*)p.out + ((size_t)b * p.Nq + q) * p.ldo + h * D;
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * half;
                    if (d < D) {
                        h16x4 v4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v4[e] = (h16)(o[db][4 * g + e] * inv);
                        *reinterpret_cast<h16x4*>(orow + d) = v4;
                    }
                }
        }
        qf = qn;
    }
}

#endif

// Round 6 (profiles/r06_experiments.txt item 4): the loop was bound by its vector instructions (493 per 32-query block against 21
// MFMAs; vector pipe 0.46 busy, matrix pipe 0.14), so
//  - Q is NOT pre-scaled (24 multiplies + conversions per block): the softmax computes exp2(fma(s, c, -m c)) as packed v_pk_fma_f32;
//  - K80 (64 < Nk <= 80: the 77-key prompt context): the ragged third key block takes its -inf mask as eight loop-invariant
//    addends instead of 48 compares + selects per block (whose 48 lane masks hipcc kept in VGPR lanes: a v_readlane each), and
//    its dead upper half (keys 80-95) is never exponentiated or multiplied;
//  - a workgroup walks query blocks ACROSS the batch elements that share its K / V (b = bkv, bkv + Bkv, ...: the CFG halves of every
//    image read the same prompt), so the 256-query level of the U-Net (d = 160: two blocks per batch element) amortises its staging
//    over eight blocks like the others; block order [bkv][chunk][head]: the heads of a row stay neighbours on one XCD.
template <int D, bool K80>
__global__ __launch_bounds__(256, 2) void attn_short_kernel(const AttnArgs p, const float scale_log2, const int qit_) {
#ifdef DSIM_DEVTOOLS
    const int qit = qit_ & 255, abl = qit_ >> 8;        // kbench ablations: 1 = no output stores, 2 = no Q prefetch loads (timing only)
#else
    const int qit = qit_;
    constexpr int abl = 0;
#endif
    typedef h16 T;
    typedef ACfg<T, D> C;
    constexpr int KR = 96;                                   // key rows held (three 32-row MFMA blocks)
    constexpr int CPRD = D / C::VEC;                         // real 16-byte chunks per row
    constexpr int VOFF = KR * C::RS;                         // V image behind the K image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int nqb = (p.Nq + 127) / 128;                      // query blocks per batch element
    const int nbmax = (p.B + p.Bkv - 1) / p.Bkv;             // batch elements per K / V
    const int nch = (nbmax * nqb + qit - 1) / qit;           // chunks of qit blocks per (K / V, head)
    int bid = blockIdx.x;
    if (p.xcd_remap) {          // every XCD a contiguous run of (K / V, chunk, head) items -- the heads of a row share cache lines
        const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, r = nwg & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
    }
    const int h = bid % p.H, t0 = bid / p.H, ch = t0 % nch, bkv = t0 / nch;
    const int nbg = (p.B - bkv + p.Bkv - 1) / p.Bkv;         // batch elements that read this K / V
    if (ch * qit >= nbg * nqb) return;
    const size_t kvoff = (size_t)bkv * p.Nk * p.ldk + h * D;
    const T* kb = (const T*)p.k + kvoff;
    const T* vb = (const T*)p.v + kvoff;
    // ---- stage K and V once: zero image (padding columns, rows >= Nk), then the real chunks and V's ones column ---------
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int o = tid * 16; o < KR * (C::RS + C::RSV); o += 256 * 16) *reinterpret_cast<u32x4*>(smem + o) = z;
        __syncthreads();
        for (int idx = tid; idx < p.Nk * CPRD; idx += 256) {
            const int r = idx / CPRD, c = idx - r * CPRD;
            const u32x4 kk = *reinterpret_cast<const u32x4*>(kb + (size_t)r * p.ldk + c * C::VEC);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(vb + (size_t)r * p.ldk + c * C::VEC);
            *reinterpret_cast<u32x4*>(smem + r * C::RS + c * 16) = kk;
            *reinterpret_cast<u32x4*>(smem + VOFF + r * C::RSV + c * 16) = vv;
        }
        if constexpr (C::ONES)
            for (int r = tid; r < p.Nk; r += 256) *reinterpret_cast<u32x4*>(smem + VOFF + r * C::RSV + CPRD * 16) = one_chunk<T>();
        __syncthreads();
    }
    const int i16 = lane & 15, g4 = lane >> 4;
    const char* const kfr = smem + l31 * C::RS + half * 16;
    const char* const vfr = smem + VOFF + (4 * (g4 >> 1) + (i16 >> 2)) * C::RSV + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2;
    // unscaled Q fragments of a block's 32 rows of this wave (rows past the end: the last row, never stored)
    auto load_q_raw = [&](QFrags<T, D>& f, int b, int q) {
        const int qq = q < p.Nq ? q : p.Nq - 1;
        const T* qrow = (const T*)p.q + ((size_t)b * p.Nq + qq) * p.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) {
            const int d0 = 16 * ks + 8 * half;
            if (d0 < D) f.f[ks] = *reinterpret_cast<const h16x8*>(qrow + d0);
            else zero_frag(f.f[ks]);
        }
    };
    // K80: the third key block's mask as eight loop-invariant addends (key = 64 + (r & 3) + 8 (r >> 2) + 4 half; r >= 8 is dead)
    float bias[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bias[r] = (64 + (r & 3) + 8 * (r >> 2) + 4 * half >= p.Nk) ? -INFINITY : 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 c2 = {scale_log2, scale_log2};
    const bool wide = wide_rows(p);

    int bi = (ch * qit) / nqb, qblk = ch * qit - bi * nqb;
    QFrags<T, D> qf;
    load_q_raw(qf, bi * p.Bkv + bkv, qblk * 128 + wave * 32 + l31);
    for (int it = 0; it < qit && bi < nbg; ++it) {
        const int b = bi * p.Bkv + bkv;
        const int q = qblk * 128 + wave * 32 + l31;
        if (++qblk == nqb) { qblk = 0; ++bi; }
        // the next block's Q rows fly while this block computes
        QFrags<T, D> qn;
        if (!(abl & 2)) load_q_raw(qn, (bi < nbg ? bi : nbg - 1) * p.Bkv + bkv, qblk * 128 + wave * 32 + l31);
        else qn = qf;
        if (q - l31 < p.Nq) {                               // wave-uniform: some of this wave's rows exist
            f32x16 s[3];
#pragma unroll
            for (int kbk = 0; kbk < 3; ++kbk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kbk][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < C::NKS; ++ks) {
                    h16x8 kf;
                    lload_frag(kf, kfr + kbk * 32 * C::RS + ks * 32);
                    mma(kf, qf.f[ks], s[kbk]);
                }
            }
            // exact row maximum over the (<= 96) keys: lane-local + one exchange between the halves
            float m = -INFINITY;
#pragma unroll
            for (int kbk = 0; kbk < 3; ++kbk)
#pragma unroll
                for (int r = 0; r < ((K80 && kbk == 2) ? 8 : 16); ++r) {
                    if constexpr (!K80) {
                        const int kv = kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (kv >= p.Nk) s[kbk][r] = -INFINITY;
                    } else if (kbk == 2) {
                        s[kbk][r] += bias[r];
                    }
                    m = fmaxf(m, s[kbk][r]);
                }
            m = max_halves(m);
            const float mc = -m * scale_log2;
            const f32x2 m2 = {mc, mc};
            float psum = 0.f;
#pragma unroll
            for (int kbk = 0; kbk < 3; ++kbk)
#pragma unroll
                for (int r = 0; r < ((K80 && kbk == 2) ? 8 : 16); r += 2) {
                    const f32x2 e = __builtin_elementwise_fma((f32x2){s[kbk][r], s[kbk][r + 1]}, c2, m2);
                    s[kbk][r] = __builtin_amdgcn_exp2f(e[0]);
                    s[kbk][r + 1] = __builtin_amdgcn_exp2f(e[1]);
                    if constexpr (!C::ONES) psum += s[kbk][r] + s[kbk][r + 1];
                }
            f32x16 o[C::NDB];
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
            for (int kbk = 0; kbk < 3; ++kbk)
#pragma unroll
                for (int s2 = 0; s2 < ((K80 && kbk == 2) ? 1 : 2); ++s2) {
                    h16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (h16)s[kbk][8 * s2 + e];
#pragma unroll
                    for (int db = 0; db < C::NDB; ++db) {
                        const char* pa = vfr + (kbk * 32 + 16 * s2) * C::RSV + db * 64;
                        const h16x4 lo = h16_ds_read_tr16_b64((pa));
                        const h16x4 hi = h16_ds_read_tr16_b64((pa + 8 * C::RSV));
                        h16x8 vf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                        o[db] = H16_MFMA_32x32x16(vf, pf, o[db], 0, 0, 0);
                    }
                }
            float l_tot;
            if constexpr (C::ONES) {
                constexpr int RB = D / 32, RR = D % 32;
                constexpr int RH = (RR >> 2) & 1, REG = (RR & 3) + 4 * (RR >> 3);
                const float mine = o[RB][REG];
                const float other = __shfl_xor(mine, 32);
                l_tot = (half == RH) ? mine : other;
            } else {
                l_tot = psum + __shfl_xor(psum, 32);
            }
            const float inv = 1.0f / l_tot;
            if (q < p.Nq && !(abl & 1)) {
                T* orow = (T*)p.out + ((size_t)b * p.Nq + q) * p.ldo + h * D;
#ifdef DSIM_DEVTOOLS
                if (abl & 4) {      // kbench: the same bytes as fully coalesced 1 KB wave stores at wrong addresses (timing only)
                    char* base = (char*)p.out + (((size_t)b * p.Nq + (q - l31)) * p.ldo) * 2 + (size_t)h * 32 * D * 2;
                    int i = 0;
#pragma unroll
                    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp)
                            if (db * 32 + 16 * gp < D) {
                                u32x4 w;
#pragma unroll
                                for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(o[db][8 * gp + e] * inv);
                                if (i * 1024 + lane * 16 < 32 * D * 2) *reinterpret_cast<u32x4*>(base + i * 1024 + lane * 16) = w;
                                ++i;
                            }
                } else
#endif
                store_o_rows<D>(orow, half, wide, [&](int db, int r) { return o[db][r] * inv; });
            }
        }
        qf = qn;
    }
}

// ---- fused score tail ----------------------------------------------------------------------
// grid (ceil(N/128), B*H, n_pairs*2); partial layout [pair][dir][bh][qtile][4] f32
// 16-bit modes (round 5): both SDPA outputs are rounded to the compute dtype before the products -- torch's SDPA returns
// tensors of the pipeline dtype and the reference's cosine / mse consume those (diffsim.py:177-190); the products and sums stay
// f32 per workgroup and f64 across them.  The self-attention's output then waits for the cross-attention as packed 16-bit
// pairs (40 registers at d = 160 instead of 80), which brings d = 160 from 426 registers (one workgroup per CU) under 256: two
// workgroups per CU.  The f32 parity mode keeps both outputs in f32.
template <typename T, int D>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void pair_tail_kernel(const T* __restrict__ qg, const T* __restrict__ kg,
                                                        const T* __restrict__ vg, const int32_t* __restrict__ idx_a,
                                                        const int32_t* __restrict__ idx_b, int B, int H, int N,
                                                        float scale_log2, int mse, float* __restrict__ part) {
    typedef ACfg<T, D> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int pair = blockIdx.z >> 1, dir = blockIdx.z & 1;
    const int ia = idx_a[pair], ib = idx_b[pair];
    const int iq = dir ? ib : ia;        // query image (also the "self" keys/values)
    const int ix = dir ? ia : ib;        // the other image ("cross" keys/values)
    const int ld = H * D;
    const size_t img = (size_t)B * N * ld;
    const int q = blockIdx.x * 128 + wave * 32 + l31;
    const int qc = q < N ? q : N - 1;
    const size_t boff = (size_t)b * N * ld + h * D;
    QFrags<T, D> qf;
    load_q<T, D>(qf, qg + iq * img + boff + (size_t)qc * ld, half, scale_log2);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if constexpr (sizeof(T) == 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        h16x2 osp[C::NDB][8];           // the self-attention's output, rounded to the compute dtype, two values per register
        {
            OAcc<T, D> osa;
            attend<T, D>(qf, kg + iq * img + boff, vg + iq * img + boff, ld, N, smem, osa);
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    osp[db][r >> 1] = __builtin_convertvector((f32x2){osa.b[db][r], osa.b[db][r + 1]}, h16x2);
                    asm volatile("" : "+v"(osp[db][r >> 1]));          // (pinned: the f32 accumulators die here, before the second attention)
                }
        }
        OAcc<T, D> oxa;
        attend<T, D>(qf, kg + ix * img + boff, vg + ix * img + boff, ld, N, smem, oxa);
        if (q < N) {
#pragma unroll
            for (int db = 0; db < C::NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const h16x2 xp = __builtin_convertvector((f32x2){oxa.b[db][r], oxa.b[db][r + 1]}, h16x2);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int rr = r + e, d = db * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
                        if (d < D) {
                            const float x = (float)xp[e], y = (float)osp[db][r >> 1][e];
                            if (mse) { const float df = x - y; s0 = fmaf(df, df, s0); }
                            else { s0 = fmaf(x, y, s0); s1 = fmaf(x, x, s1); s2 = fmaf(y, y, s2); }
                        }
                    }
                }
        }
    } else {
    OAcc<T, D> osa, oxa;
    attend<T, D>(qf, kg + iq * img + boff, vg + iq * img + boff, ld, N, smem, osa);
    attend<T, D>(qf, kg + ix * img + boff, vg + ix * img + boff, ld, N, smem, oxa);
    auto& os = osa.b;
    auto& ox = oxa.b;
    if (q < N) {
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (d < D) {
                    const float x = ox[db][r], y = os[db][r];
                    if (mse) { const float df = x - y; s0 = fmaf(df, df, s0); }
                    else { s0 = fmaf(x, y, s0); s1 = fmaf(x, x, s1); s2 = fmaf(y, y, s2); }
                }
            }
    }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + ((((size_t)pair * 2 + dir) * gridDim.y + bh) * gridDim.x + blockIdx.x) * 4;
        o[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        o[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        o[2] = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
        o[3] = 0.f;
    }
}

// one thread per pair: fixed-order f64 fold of the partials, then cosine / mse and the mean of
// the two directions (diffsim.py:187-197; F.cosine_similarity eps = 1e-8)
__global__ void pair_finish_kernel(const float* __restrict__ part, int n_pairs, int nblk, int mse, double count,
                                   float* __restrict__ out, int32_t* __restrict__ status) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    double res = 0.0;
    for (int dir = 0; dir < 2; ++dir) {
        double a = 0.0, x2 = 0.0, y2 = 0.0;
        const float* o = part + ((size_t)p * 2 + dir) * nblk * 4;
        for (int i = 0; i < nblk; ++i) { a += o[4 * i]; x2 += o[4 * i + 1]; y2 += o[4 * i + 2]; }
        if (mse) res += a / count;
        else {
            const double nx = sqrt(x2), ny = sqrt(y2);
            res += a / (fmax(nx, 1e-8) * fmax(ny, 1e-8));
        }
    }
    const float sc = (float)(res * 0.5);
    out[p] = sc;
    // NaN guard: non-finite features surface here as a non-finite score; report them per pair
    if (status) status[p] = (sc - sc == 0.0f) ? 0 : 1;
}

inline float scale_log2_of(int D) { return (1.0f / sqrtf((float)D)) * 1.4426950408889634f; }

#ifdef DSIM_DEVTOOLS
// kbench occupancy probe: g_attn_lds_pad KB of unused LDS on top of every tiled attention launch
#define A_LAUNCH_LDS(kern, base) ([&]() { const int l_ = (base) + g_attn_lds_pad * 1024; if (g_attn_lds_pad) (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, l_); return l_; }())
#define DSIM_SHORT_QIT(q) ((q) | (g_attn_dbg << 8))
#else
#define A_LAUNCH_LDS(kern, base) (base)
#define DSIM_SHORT_QIT(q) (q)
#endif

template <typename T, int D>
int launch_attn_d(const AttnArgs& a, hipStream_t s) {
    typedef ACfg<T, D> C;
    const dim3 grid(((a.Nq + 127) / 128) * a.H * a.B);
    // long key sequences (>= 1024 keys: the 64 x 64 and 32 x 32 self-attentions): the fixed-reference softmax -- 6 % faster at
    // 4096 keys x d = 40, 10 % at 1024 keys x d = 80, 14 % at 1024 keys x d = 64 (SDXL); short ones (cross-attention's 77 keys,
    // the 16 x 16 level: 0.179 -> 0.195 ms at 256 keys x d = 160) keep the exact running maximum -- there the end-of-block check
    // costs more than the skipped maxima save
    if constexpr (sizeof(T) == 2 && (D == 40 || D == 64 || D == 80 || D == 160)) {
        // the prompt context of the cross-attentions (77 keys): keys resident in LDS, several query blocks per workgroup
        // (round 6, interleaved A/B against the tiled kernel at 64 pairs: d = 40 0.492 -> 0.387 ms, d = 80 0.221 -> 0.199, d = 160 at
        //  256 queries 0.112 -> 0.091, SDXL's d = 64 at 4096 / 1024 queries 0.736 -> 0.652 / 0.397 -> 0.381)
        if (a.Nk <= 96 && g_attn_short) {
            // TWO workgroups per CU whatever the K / V images need: the kernel is bound by the cache-line operations of its Q rows in
            // and O rows out, and more resident workgroups only thrash that path (64 pairs, ms at 4 | 3 | 2 | 1 workgroups per CU:
            // d = 40  0.457 | 0.426 | 0.388 | 0.447;  d = 80  0.193 | 0.181 | 0.180 | 0.191: profiles/r06_experiments.txt 4e)
            constexpr int LDSK = 96 * (C::RS + C::RSV);
            constexpr int LDSS = LDSK < 56 * 1024 ? 56 * 1024 : LDSK;
#ifdef DSIM_DEVTOOLS
            if (g_attn_short == 2) {
                static DeviceOnce onces;
                auto kern = attn_short_v1_kernel<D>;
                CK_ONCE(onces, kern, LDSK);
                const int nqb = (a.Nq + 127) / 128;
                int qit = 8;
                while (qit > 1 && (long)((nqb + qit - 1) / qit) * a.H * a.B < 8L * cu_count()) qit >>= 1;
                hipLaunchKernelGGL(kern, dim3(((nqb + qit - 1) / qit) * a.H * a.B), dim3(256), LDSK, s, a, scale_log2_of(D), qit);
                DSIM_HIP_CHECK(hipGetLastError());
                return DSIM_OK;
            }
#endif
            // query blocks per workgroup: as many as keep >= 8 workgroups per CU in the grid (at most 8); the 256-register d = 160
            // instantiation (two workgroups per CU) is content with one full round.  A workgroup's blocks continue across the batch
            // elements that share its K / V.
            const int nqb = (a.Nq + 127) / 128, nvb = ((a.B + a.Bkv - 1) / a.Bkv) * nqb;
            const long want = (D > 80 ? 2L : 8L) * cu_count();
            int qit = 8;
            while (qit > 1 && (long)((nvb + qit - 1) / qit) * a.H * a.Bkv < want) qit >>= 1;
            const int nch = (nvb + qit - 1) / qit;
            const dim3 g(nch * a.H * a.Bkv);
            if (a.Nk > 64 && a.Nk <= 80) {
                static DeviceOnce onces;
                auto kern = attn_short_kernel<D, true>;
                CK_ONCE(onces, kern, LDSS);
                hipLaunchKernelGGL(kern, g, dim3(256), LDSS, s, a, scale_log2_of(D), DSIM_SHORT_QIT(qit));
            } else {
                static DeviceOnce onces;
                auto kern = attn_short_kernel<D, false>;
                CK_ONCE(onces, kern, LDSS);
                hipLaunchKernelGGL(kern, g, dim3(256), LDSS, s, a, scale_log2_of(D), DSIM_SHORT_QIT(qit));
            }
            DSIM_HIP_CHECK(hipGetLastError());
            return DSIM_OK;
        }
    }
    if constexpr (sizeof(T) == 2 && D == 40) {
        // two query blocks per wave where SD1.5's 4096-key level lives (d = 64 would spill: its K fragments and staging are wider)
        if (a.Nk >= 2048 && a.Nk % KT == 0 && g_attn_q2) {
            constexpr int LDS3 = 3 * 2 * C::TILE;           // the 3-deep tile ring
#ifdef DSIM_DEVTOOLS
            switch (g_attn_dbg) {
#define X(d) case d: { static DeviceOnce o; auto k = attn_long_kernel<D, d>; CK_ONCE(o, k, LDS3); hipLaunchKernelGGL(k, dim3(((a.Nq + 255) / 256) * a.H * a.B), dim3(256), LDS3, s, a, scale_log2_of(D)); DSIM_HIP_CHECK(hipGetLastError()); return DSIM_OK; }
                X(8) X(32) X(64) X(96)
#undef X
                default: break;
            }
#endif
            static DeviceOnce once2;
            auto kern = attn_long_kernel<D, 0>;
            CK_ONCE(once2, kern, LDS3);
            hipLaunchKernelGGL(kern, dim3(((a.Nq + 255) / 256) * a.H * a.B), dim3(256), A_LAUNCH_LDS(kern, LDS3), s, a, scale_log2_of(D));
            DSIM_HIP_CHECK(hipGetLastError());
            return DSIM_OK;
        }
    }
    if constexpr (sizeof(T) == 2 && D == 64) {
        // two query blocks per wave (attend2; d = 80 would spill 82-175 registers): halves the LDS fragment reads that bound attn_kernel; worth it from 256 queries up
        if (a.Nk > 96 && a.Nq >= 256 && g_attn_q2) {
            const dim3 grid2(((a.Nq + 255) / 256) * a.H * a.B);
            if (a.Nk >= g_attn_fast_min) {
                static DeviceOnce o1;
                auto k = attn_q2_kernel<D, true>;
                CK_ONCE(o1, k, C::LDS);
                hipLaunchKernelGGL(k, grid2, dim3(256), A_LAUNCH_LDS(k, C::LDS), s, a, scale_log2_of(D));
            } else {
                static DeviceOnce o2;
                auto k = attn_q2_kernel<D, false>;
                CK_ONCE(o2, k, C::LDS);
                hipLaunchKernelGGL(k, grid2, dim3(256), A_LAUNCH_LDS(k, C::LDS), s, a, scale_log2_of(D));
            }
            DSIM_HIP_CHECK(hipGetLastError());
            return DSIM_OK;
        }
    }
    if (sizeof(T) == 2 && a.Nk >= g_attn_fast_min) {
        static DeviceOnce oncef;
        auto kern = attn_kernel<T, D, true>;
        CK_ONCE(oncef, kern, C::LDS);
        hipLaunchKernelGGL(kern, grid, dim3(256), A_LAUNCH_LDS(kern, C::LDS), s, a, scale_log2_of(D));
    } else {
        static DeviceOnce once;
        auto kern = attn_kernel<T, D, false>;
        CK_ONCE(once, kern, C::LDS);
        hipLaunchKernelGGL(kern, grid, dim3(256), A_LAUNCH_LDS(kern, C::LDS), s, a, scale_log2_of(D));
    }
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

// head dims of the supported graphs: SD1.5 40/80/160, SDXL 64, DiT-XL/2 72, test configs 16/32/64
#define DSIM_FOR_EACH_D(X) X(16) X(32) X(40) X(64) X(72) X(80) X(160)

template <typename T>
int launch_attn_t(const AttnArgs& a, hipStream_t s) {
    switch (a.D) {
#define X(d) case d: return launch_attn_d<T, d>(a, s);
        DSIM_FOR_EACH_D(X)
#undef X
        default: return DSIM_ERR_INVALID;
    }
}

template <typename T, int D>
int launch_tail_d(const void* q, const void* k, const void* v, const int32_t* ia, const int32_t* ib, int n_pairs,
                  int B, int H, int N, int mse, float* out, void* scratch, hipStream_t s, int32_t* status) {
    typedef ACfg<T, D> C;
    static DeviceOnce once;
    auto kern = pair_tail_kernel<T, D>;
    CK_ONCE(once, kern, C::LDS);
    const int qt = (N + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(qt, B * H, n_pairs * 2), dim3(256), C::LDS, s, (const T*)q, (const T*)k,
                       (const T*)v, ia, ib, B, H, N, scale_log2_of(D), mse, (float*)scratch);
    hipLaunchKernelGGL(pair_finish_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, s, (const float*)scratch, n_pairs,
                       qt * B * H, mse, (double)B * H * N * D, out, status);
    DSIM_HIP_CHECK(hipGetLastError());
    return DSIM_OK;
}

template <typename T>
int launch_tail_t(const void* q, const void* k, const void* v, const int32_t* ia, const int32_t* ib, int n_pairs,
                  int B, int H, int N, int D, int mse, float* out, void* scratch, hipStream_t s, int32_t* status) {
    switch (D) {
#define X(d) case d: return launch_tail_d<T, d>(q, k, v, ia, ib, n_pairs, B, H, N, mse, out, scratch, s, status);
        DSIM_FOR_EACH_D(X)
#undef X
        default: return DSIM_ERR_INVALID;
    }
}

}  // namespace

#ifdef DSIM_DEVTOOLS
int g_attn_q2 = 1;
int g_attn_dbg = 0;
int g_attn_lds_pad = 0;
int g_norm_lds_pad = 0;
int g_attn_short = 1;
int g_attn_fast_min = 1024;
int g_tail160 = 1;
int g_sdpa160 = 1;
#endif

// Which kernel launch_attention picks for this problem, as the suffix of the profile family name (bench.py maps family names to
// the symbols rocprofv3 prints): "_short" attn_short_kernel (keys resident in LDS), "_long" attn_long_kernel (two query blocks
// per wave, pipelined), "_q2" / "_q2fast" attn_q2_kernel (two query blocks per wave sharing every fragment read; exact / fixed-reference
// softmax), "_fast" attn_kernel with the fixed-reference softmax, "_p160" sdpa160_kernel (attn160.hip: 256 x 256 tokens at d = 160 on the
// persistent core), "" attn_kernel with the exact running maximum.
// Mirrors launch_attn_d's conditions (the development switches are 1 in the product).
const char* attention_kernel_kind(const AttnArgs& a, int dtype) {
    if (dtype == DSIM_F32) return "";
    if (g_sdpa160 && sdpa160_applies(a)) return "_p160";
    if ((a.D == 40 || a.D == 64 || a.D == 80 || a.D == 160) && a.Nk <= 96) return "_short";
    if (a.D == 40 && a.Nk >= 2048 && a.Nk % KT == 0) return "_long";
    if (a.D == 64 && a.Nk > 96 && a.Nq >= 256) return a.Nk >= g_attn_fast_min ? "_q2fast" : "_q2";
    if (a.Nk >= g_attn_fast_min) return "_fast";
    return "";
}

int launch_attention(const AttnArgs& a, int dtype, hipStream_t s) {
    const int vec = dtype == DSIM_F32 ? 4 : 8;
    if (a.D % 8 || a.ldq % vec || a.ldk % vec || a.ldo % 4 || a.Nk < 1 || a.Nq < 1 || a.Bkv < 1)
        return DSIM_ERR_INVALID;
    if (dtype == DSIM_H16) {
        if (g_sdpa160 && sdpa160_applies(a)) return launch_sdpa160(a, s);      // 256 x 256 tokens at d = 160: the persistent core (attn160.hip)
        return launch_attn_t<h16>(a, s);
    }
#ifndef DSIM_H16_IS_F16
    if (dtype == DSIM_F32) return launch_attn_t<float>(a, s);
#ifdef DSIM_HAS_F16_TWINS
    if (dtype == DSIM_F16) return launch_attention_f16(a, dtype, s);
#endif
#endif
    return DSIM_ERR_INVALID;
}

size_t pair_score_scratch_bytes(int n_pairs, int B, int H, int N, int D) {
    const size_t tiled = (size_t)n_pairs * 2 * B * H * ((N + 127) / 128) * 4 * sizeof(float);
    if (pair_score160_applies(N, D, DSIM_H16)) {          // (dtype-blind: the 16-bit modes' persistent kernel needs the larger workspace)
        const size_t pers = pair_score160_scratch_bytes(n_pairs, B, H);
        return pers > tiled ? pers : tiled;
    }
    return tiled;
}

int launch_pair_score(const void* q, const void* k, const void* v, const int32_t* ia, const int32_t* ib,
                      int n_pairs, int B, int H, int N, int D, int dtype, int similarity, float* out, void* scratch,
                      size_t scratch_bytes, hipStream_t s, int32_t* status) {
    if (n_pairs <= 0 || D % 8 || N < 1) return DSIM_ERR_INVALID;
    if (scratch_bytes < pair_score_scratch_bytes(n_pairs, B, H, N, D)) return DSIM_ERR_WORKSPACE;
    if (n_pairs * 2 > 65535) return DSIM_ERR_INVALID;
    if (dtype == DSIM_H16) {
        if (g_tail160 && pair_score160_applies(N, D, DSIM_H16))
            return launch_pair_score160(q, k, v, ia, ib, n_pairs, B, H, similarity, out, scratch, scratch_bytes, s, status);
        return launch_tail_t<h16>(q, k, v, ia, ib, n_pairs, B, H, N, D, similarity, out, scratch, s, status);
    }
#ifndef DSIM_H16_IS_F16
    if (dtype == DSIM_F32) return launch_tail_t<float>(q, k, v, ia, ib, n_pairs, B, H, N, D, similarity, out, scratch, s, status);
#ifdef DSIM_HAS_F16_TWINS
    if (dtype == DSIM_F16)
        return launch_pair_score_f16(q, k, v, ia, ib, n_pairs, B, H, N, D, dtype, similarity, out, scratch, scratch_bytes, s, status);
#endif
#endif
    return DSIM_ERR_INVALID;
}

}  // namespace dsim
