// Internal shared declarations for libdiffsim_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/diffsim_amd.h"

// ---- second compilation (h16 = fp16): the entry points of gemm / rowres / norm / attention .hip under an _f16 suffix ----------
#ifdef DSIM_H16_IS_F16
#define launch_gemm launch_gemm_f16
#define gemm_tile_choice gemm_tile_choice_f16
#define gemm_launch_tile gemm_launch_tile_f16
#define gemm_band_width gemm_band_width_f16
#define gemm_fill_extents gemm_fill_extents_f16
#define gemm_skinny_applies gemm_skinny_applies_f16
#define gemm_skinny_tile gemm_skinny_tile_f16
#define launch_gemm_skinny launch_gemm_skinny_f16
#define cu_count cu_count_f16
#define rowlin_stream_bytes rowlin_stream_bytes_f16
#define pack_rowlin_stream pack_rowlin_stream_f16
#define launch_rowlin launch_rowlin_f16
#define groupnorm_scratch_bytes groupnorm_scratch_bytes_f16
#define groupnorm_passes groupnorm_passes_f16
#define launch_groupnorm launch_groupnorm_f16
#define launch_groupnorm_pre launch_groupnorm_pre_f16
#define launch_layernorm launch_layernorm_f16
#define launch_layernorm_mod launch_layernorm_mod_f16
#define launch_softmax_rows launch_softmax_rows_f16
#define launch_attention launch_attention_f16
#define attention_kernel_kind attention_kernel_kind_f16
#define pair_score_scratch_bytes pair_score_scratch_bytes_f16
#define launch_pair_score launch_pair_score_f16
#define pair_score160_applies pair_score160_applies_f16
#define pair_score160_scratch_bytes pair_score160_scratch_bytes_f16
#define launch_pair_score160 launch_pair_score160_f16
#define sdpa160_applies sdpa160_applies_f16
#define launch_sdpa160 launch_sdpa160_f16
#define ff_stream_bytes ff_stream_bytes_f16
#define pack_ff_stream pack_ff_stream_f16
#define launch_ff_fused launch_ff_fused_f16
#endif

namespace dsim {

// The 16-bit compute type.  The kernel sources (gemm / rowres / norm / attention .hip) are written against ONE 16-bit type,
// h16, and are compiled twice (diffsim_amd/build.py): once with h16 = bf16 (compute dtype DSIM_BF16, the headline mode) and
// once with -DDSIM_H16_IS_F16, h16 = IEEE fp16 (DSIM_F16: the arithmetic type the reference's drivers run in,
// /root/reference/cute_main.py:31, diffsim/diffsim.py:82).  The fp16 objects carry the same entry points under an _f16
// suffix (the #define block at the end of this header); the bf16 objects own the plain names and forward DSIM_F16 calls.
// v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 take the same cycles as the bf16 forms, so tiles and schedules are shared.
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
#ifdef DSIM_H16_IS_F16
typedef _Float16 h16;
#define DSIM_H16 DSIM_F16
#define DSIM_H16_ONE_BITS 0x3C00u
// fixed-reference softmax (attention.hip attend<FAST>): P = exp2(s - m0) is stored in the 16-bit type; fp16 tops out at 65504, so
// a row whose sum reaches 3e4 (a single P near the limit, or thousands of keys a few units above tile 0's maximum) takes the exact
// running-maximum form instead (in bf16 the bound is the f32 exponent range)
#define DSIM_H16_LSUM_MAX 3.0e4f
#define H16_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define H16_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 h16;
#define DSIM_H16 DSIM_BF16
#define DSIM_H16_ONE_BITS 0x3F80u
#define DSIM_H16_LSUM_MAX 1e30f
#define H16_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define H16_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(2))) h16 h16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#ifdef __HIPCC__
// ds_read_b64_tr_b16 of the 16-bit compute type: p is a (generic) pointer into LDS
__device__ __forceinline__ h16x4 h16_ds_read_tr16_b64(const char* p) {
#ifdef DSIM_H16_IS_F16
    typedef __fp16 fp16v4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    return __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16v4*)(p)));
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) h16x4*)(p));
#endif
}
// GELU (erf form, F.gelu's default) for h16 outputs: x * sigmoid(x (a + b u + c u^2)), u = min(x^2, 64), with -log2(e) folded
// into the constants (the clamp keeps the odd quintic monotone).  |error| <= 2.6e-5 absolute, <= 0.3 h16 ulp of the result:
// one v_exp, one v_rcp and 7 plain VALU operations against 16 + 2 for the erf polynomial the fp32 parity mode keeps.  Used by
// every h16 GEGLU (the GEMM epilogue and the fused feed-forward), so the fused and unfused chains round alike.
__device__ __forceinline__ float gelu_fast(float x) {
    const float u = fminf(x * x, 64.0f);
    const float t = x * fmaf(u, fmaf(u, 1.01426306e-3f, -0.106775724f), -2.30112134f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
#endif

#define DSIM_HIP_CHECK(expr)                                   \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) return DSIM_ERR_HIP;             \
    } while (0)

static inline size_t dtype_size(int dt) { return dt == DSIM_F32 ? 4 : 2; }

// ---------------------------------------------------------------------------------------------
// implicit-GEMM (linear / 1x1 conv / 3x3 conv) -- gemm.hip
//   out[m][n] = epi( sum_k A(m,k) * W[n][k] + bias[n] )
// A is gathered on the fly from token-major activations:
//   LINEAR : A(m,k) = k < C0 ? A0[m*C0 + k] : A1[m*C1 + (k-C0)]      (channel concat of two sources)
//   CONV3  : k = tap*C0 + c ; A(m,k) = X[b][iy][ix][c], zero outside, optional stride 2 /
//            nearest-2x upsample folded into the index
// ---------------------------------------------------------------------------------------------
enum GemmMode { GEMM_LINEAR = 0, GEMM_CONV3 = 1, GEMM_CONV3P = 2 };   // CONV3P: gemm_kernel's instantiation for power-of-two output maps (never in GemmArgs.mode)
enum GemmEpi { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_GEGLU = 2 };

struct GemmArgs {
    const void* A0 = nullptr;
    const void* A1 = nullptr;
    int C0 = 0, C1 = 0;
    int mode = GEMM_LINEAR;
    int Hin = 0, Win = 0, Hout = 0, Wout = 0;   // CONV3 geometry (Hin/Win = stored input size)
    int stride = 1, ups = 0;
    int pad = 1;                                // 1: symmetric padding; 0: VAE downsample (pad right/bottom only)
    int lwo = -1, lhw = -1;                     // filled by launch_gemm: log2(Wout), log2(Hout * Wout) when both are powers of two, else -1
    int M = 0, N = 0, K = 0;                    // N counts packed weight rows (2x out cols for GEGLU)
    const void* W = nullptr;                    // packed [N][K], compute dtype
    const float* bias = nullptr;                // [N] f32 (packed order) or null
    const float* bias2 = nullptr;               // optional: bias of ODD batch elements (m / rows_per_batch) & 1 -- SDXL's
    int rows_per_batch = 0;                     //   time embedding differs between the uncond/cond CFG halves
    int act = 0;                                // 1: tanh-GELU after the bias (DiT Mlp.fc1)
    const float* gate = nullptr;                // optional per-column scale applied before the residual add (DiT adaLN
    const float* gate2 = nullptr;               //   gates); gate2 = the ODD batch elements' vector
    int epi = EPI_NONE;
    int geglu_blk = 32;                         // EPI_GEGLU: rows per alternating [h | g] block of the packed weight (geglu_block_rows())
    const void* residual = nullptr;             // [M][ldo]
    void* out = nullptr;
    int ldo = 0;
    int out_split = 0;                          // > 0 (plain linear only, % 320 == 0): output columns [j*split, (j+1)*split) go to the
    long long out_split_stride = 0;             //   tensor at out + j * out_split_stride bytes, each [M][ldo] (the tapped q | k | v)
    int force_big = 0;                          // 1: 256-row tiles whatever the tile count (a caller that needs the SAME tiles at every batch size:
                                                //    the VAE's 128 x 128 level asks for epilogue statistics, whose partial sums follow the tile shape)
    int wb_rows = 0;                            // optional (GEMM_LINEAR): batched weights -- rows [i * wb_rows, (i + 1) * wb_rows) multiply the
    unsigned wb_stride = 0;                     //   [N][K] matrix at W + i * wb_stride bytes (the VAE's per-image q k^T and P v); M % wb_rows == 0
    float* gn_part = nullptr;                   // optional (16-bit 3x3 convs on 256 x 128 / 256 x 256 tiles): GroupNorm statistics of the OUTPUT from
    int gn_hw = 0;                              //   the epilogue: [image][gn_hw / 64][N / 4][2] f32 (sum, sum of squares) per (wave's 64 rows, 4-channel
                                                //   quad); gn_hw = rows per image (% 256 == 0).  launch_groupnorm_pre folds them.
    const void* zero_page = nullptr;            // >= 16 zero bytes (kept for ABI stability; padding now comes from OOB buffer reads)
    unsigned a0_bytes = 0, a1_bytes = 0, w_bytes = 0, out_bytes = 0;   // filled by launch_gemm: operand extents for the buffer descriptors
#ifdef DSIM_DEVTOOLS
    int exp = 0;                                // kernel experiments (tools/kbench)
    unsigned long long* stamps = nullptr;       // -DDSIM_STAMPS builds: per-phase cycle sums of gemm_kernel (7 words)
#endif
};
int launch_gemm(const GemmArgs& a, int dtype, hipStream_t s);
// Rows per alternating block of a GEGLU-interleaved weight with N packed rows (= 8C): 16 where the 320 / 160-column GEMM tiles
// divide N (their waves hold 160 or 80 packed rows: five or ten 16-row accumulator tiles, an odd count of 32-row blocks); 32 for
// the 320-channel blocks, whose weights the fused feed-forward streams (32 x 32 MFMAs), and for widths the 256 / 128-column tiles serve.
inline int geglu_block_rows(int N) { return (N % 320 == 0 && N != 8 * 320) ? 16 : 32; }
// Development switches (kernel A/B in tools/kbench, environment overrides): they exist only in -DDSIM_DEVTOOLS builds
// (tools/build_kbench.py); the product library compiles them away as constants.
#ifdef DSIM_DEVTOOLS
extern int g_gemm_persistent;   // 0 = one tile per workgroup
extern int g_force_bm;          // 0 = heuristic; 128/256 force the row tile
extern int g_gemm_exp;          // experiment mask passed to gemm_kernel
extern int g_gemm_skinny;       // 0 = small problems through gemm_kernel as well
extern unsigned long long* g_gemm_stamps;   // device buffer of gemm_kernel's phase stamps (-DDSIM_STAMPS builds)
extern int g_skinny_tile;       // 0 = heuristic, else (bm << 8) | bn
extern int g_gn_onepass;        // DSIM_GN_ONEPASS
extern int g_ln_rows;           // DSIM_LN_ROWS
extern int g_prep8;             // DSIM_PREP8
extern int g_attn_q2;           // 0 = long-key attention with one query block per wave (attn_kernel)
extern int g_attn_dbg;          // ablation mask of attn_long_kernel
extern int g_norm_lds_pad;      // kbench occupancy probe: KB of unused LDS per GroupNorm workgroup
extern int g_attn_lds_pad;      // kbench occupancy probe: KB of unused LDS added to the tiled attention launches
extern int g_attn_short;        // 0 = short key sequences through attn_kernel
extern int g_attn_fast_min;     // fewest keys that take the fixed-reference softmax of attn_kernel
extern int g_tail160;           // 0 = the 256-token d = 160 score tail through pair_tail_kernel
extern int g_sdpa160;           // 0 = the 256-token d = 160 self-attention through attn_kernel
extern float* g_tail160_dbg;    // kbench: device buffer for the first unit's two attention outputs
extern int g_tail160_exp;       // kbench: experiment mask of pair_tail160_kernel
extern int g_ff_dbg;            // ablation mask of the fused feed-forward kernel (rowres.hip)
extern int g_rl_dbg;            // ablation mask of the row-resident Linear kernel (rowres.hip)
extern int g_rl_wpc;            // rowlin_kernel's persistent workgroups per CU (kbench occupancy probe)
extern int g_ff_stagger;        // its wave de-phasing, in s_nop 7 units per wave index
#else
constexpr int g_gemm_skinny = 1, g_gemm_persistent = 1, g_force_bm = 0, g_gn_onepass = 1, g_ln_rows = 1, g_prep8 = 1, g_attn_q2 = 1, g_attn_short = 1, g_attn_fast_min = 1024, g_tail160 = 1, g_sdpa160 = 1;
#endif
int gemm_fill_extents(GemmArgs& g, size_t es);                       // operand byte extents for the buffer descriptors
bool gemm_skinny_applies(const GemmArgs& a);                         // small-batch kernel (gemm_skinny.hip): same arithmetic, deep ring
int launch_gemm_skinny(const GemmArgs& g /*extents filled*/, hipStream_t s);
void gemm_skinny_tile(const GemmArgs& a, int* bm, int* bn);          // its tile for this problem
int gemm_band_width(int tilesM, int tilesN, size_t w_tile_bytes);   // tile-order band width (L2 reuse of the weight tiles)
void gemm_tile_choice(const GemmArgs& a, int* bm, int* bn);   // the tile the problem's shape asks for
void gemm_launch_tile(const GemmArgs& a, int dtype, int* bm, int* bn);   // ... and the instantiation launch_gemm picks for it (dtype: DSIM_F32 or a 16-bit one)

// Can launch_gemm take GemmArgs.gn_part for this problem?  (16-bit 3x3 conv on a power-of-two output map whose tile is the 256-row
// one with 128 or 256 columns, N a multiple of it, whole images per 256 rows.)  The executors ask with the geometry of ONE image:
// where a single image already fills the chip's tiles, every batch size runs the same tiles and the statistics are batch-invariant.
inline bool gemm_gn_stats_tile(const GemmArgs& a, int dtype) {
    if (dtype == DSIM_F32 || a.mode != GEMM_CONV3 || a.epi == EPI_GEGLU || a.bias2 || a.Wout <= 0) return false;
    const int hw = a.Hout * a.Wout;
    if ((a.Wout & (a.Wout - 1)) || (hw & (hw - 1)) || hw % 256) return false;
    if (!a.force_big && gemm_skinny_applies(a)) return false;
    int bm, bn;
    gemm_launch_tile(a, dtype, &bm, &bn);
    return ((bm == 256 && (bn == 128 || bn == 256)) || (bm == 512 && bn == 128)) && a.N % bn == 0 && hw % bm == 0;
}

// weight repack kernels -- pack.hip  (src f32/h16/f16 diffusers layout -> packed compute dtype)
int pack_linear(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int K,
                int geglu_interleave, hipStream_t s);                       // [N][K] -> [N][K]; geglu_interleave: 0 or the block rows (16 / 32)
int pack_conv3(const void* src, int src_dtype, void* dst, int dst_dtype, int Cout, int Cin,
               hipStream_t s);                                              // [Co][Ci][3][3] -> [Co][9][Ci]
int pack_conv_in(const void* src, int src_dtype, float* dst, int Cout, int Cin, hipStream_t s);   // -> f32 [9*Ci][Co]
int pack_vector(const void* src, int src_dtype, float* dst, int N, int geglu_interleave,
                hipStream_t s);                                             // any -> f32
// y[n] = bias[n] + sum_k W[n][k] * act(x[k]) ; all f32, W may be f32/h16/f16 ; act: 0 id, 1 silu
int gemv_f32(const void* W, int w_dtype, const void* bias, int b_dtype, const float* x, float* y,
             int N, int K, int act, hipStream_t s);
int add_vectors_f32(const float* a, const float* b, float* out, int N, hipStream_t s);
int timestep_sincos(float* out, int dim, int t, hipStream_t s);
int sincos_values(float* out, int dim, const float* vals /*device*/, int count, hipStream_t s);

// noising + CFG duplication + conv_in (direct) -- pack.hip
//   x_t = sa*lat + sb*noise ; out[(img*2+cfg)][pix][co] for cfg in {0,1}
int prep_conv_in(const float* lat, const float* noise /*nullable*/, float sa, float sb, const float* w /*[9*Cin][Cout]*/,
                 const float* bias, void* out, int dtype, int n_img, int Cin, int S, int Cout, int dup /*1|2*/,
                 hipStream_t st);
// the VAE's 3 -> 128 conv_in at image resolution, one 64-pixel row segment per workgroup (bit-identical to prep_conv_in); gn_part
// (16-bit dtypes only, nullable): the consumer's GroupNorm statistics in the conv epilogue's format (launch_groupnorm_pre)
bool conv_in_rows_applies(int Cin, int S, int Cout);
int conv_in_rows(const float* images, const float* w /*[27][128]*/, const float* bias, void* out, int dtype, int n_img, int S,
                 float* gn_part, hipStream_t st);
int convert_f32_to(const float* src, void* dst, int dtype, size_t n, hipStream_t s);
// out[2b], out[2b+1] = in[b]: a batch element becomes its two classifier-free-guidance copies (bytes_per_elem % 16 == 0)
int dup_batch(const void* in, void* out, int n_batch, size_t bytes_per_elem, hipStream_t s);
// token-major nearest-neighbour resize [B][Hin][Win][C] -> [B][Hout][Wout][C] (row_bytes = C * element size, % 16 == 0), source
// index = min(floor(dst * (float)in / out), in - 1) as F.interpolate(mode="nearest") computes it: the explicit-size
// upsample of latent sides that are not a multiple of 2**levels (diffusers' forward_upsample_size)
int resize_nearest(const void* in, void* out, int B, int Hin, int Win, int Hout, int Wout, size_t row_bytes, hipStream_t s);

// row-resident Linear for K = 320 (optionally behind a LayerNorm): out[M][N] = LN?(x) W^T (+ bias), N % 64 == 0, N <= 960 -- rowres.hip
struct RowLinArgs {
    const void* x = nullptr;                    // [M][C] h16
    void* out = nullptr;                        // [M][N] h16
    const float* ln_g = nullptr;                // null: no LayerNorm in front
    const float* ln_b = nullptr;
    const void* stream = nullptr;               // pack_rowlin_stream output
    int M = 0, C = 0, N = 0;
    float eps = 1e-5f;
    int dtype = DSIM_BF16;                      // DSIM_BF16 or DSIM_F16
};
size_t rowlin_stream_bytes(int C, int N);       // 0: shape not covered
int pack_rowlin_stream(const void* w_packed /*[N][C] h16*/, void* stream, int C, int N, hipStream_t s);
int launch_rowlin(const RowLinArgs& a, hipStream_t s);

// uint8 HWC pixels -> process_image's normalised NCHW f32 (half: rounded through fp16); VAE posterior sample -- pack.hip
int image_preprocess(const unsigned char* hwc, float* out, int n, int H, int W, int half, hipStream_t s);
int latent_sample(const float* moments, const float* eps, float* out, int n_out, int first, int stride, int C, int hw, int eps_n,
                  float sf, int round16, hipStream_t s);

// norms -- norm.hip
size_t groupnorm_scratch_bytes(int B, int groups);
int groupnorm_passes(int C0, int C1, int HW, int groups, int dtype);    // 2 (one-pass form) or 3: algorithmic tensor passes
int launch_groupnorm(const void* x0, int C0, const void* x1, int C1, const float* gamma,
                     const float* beta, void* out, int B, int HW, int groups, float eps, int silu,
                     int dtype, void* scratch, hipStream_t s);
int launch_groupnorm_pre(const void* x, int C, const float* gamma, const float* beta, void* out, int B, int HW, int groups, float eps,
                         int silu, int dtype, void* scratch, const float* part32, int chunks, hipStream_t s);   // statistics from GemmArgs.gn_part
int launch_layernorm(const void* x, const float* gamma, const float* beta, void* out, int M, int C,
                     float eps, int dtype, hipStream_t s);
// LayerNorm without affine followed by adaLN modulate: y = LN(x) * (1 + scale[half]) + shift[half], half =
// (row / rows_per_batch) & 1 (DiT blocks; scale/shift are f32 [2][C])
int launch_layernorm_mod(const void* x, const float* scale2, const float* shift2, void* out, int M, int C,
                         int rows_per_batch, float eps, int dtype, hipStream_t s);
// out[r][:] = softmax(x[r][:] * scale) over `cols` (VAE mid-block attention; in place allowed)
int launch_softmax_rows(const void* x, void* out, int rows, int cols, float scale, int dtype, hipStream_t s);

// attention + fused score tail -- attention.hip
struct AttnArgs {
    const void* q = nullptr; int ldq = 0;     // [B][Nq] rows of ldq elements, head h at column h*D
    const void* k = nullptr; const void* v = nullptr; int ldk = 0;   // [Bkv][Nk] rows
    void* out = nullptr; int ldo = 0;
    int B = 0, Bkv = 0, H = 0, Nq = 0, Nk = 0, D = 0;
    int xcd_remap = 1;                        // 0: plain block order (micro-benchmark A/B only)
};
int launch_attention(const AttnArgs& a, int dtype, hipStream_t s);
const char* attention_kernel_kind(const AttnArgs& a, int dtype);      // "_short" / "_long" / "_fast" / "": the kernel it picks
int launch_attention_fp8(const AttnArgs& a, hipStream_t s);      // h16 in/out, e4m3 MFMAs (attention_fp8.hip)
size_t pair_score_scratch_bytes(int n_pairs, int B, int H, int N, int D);
int launch_pair_score(const void* q, const void* k, const void* v, const int32_t* idx_a,
                      const int32_t* idx_b, int n_pairs, int B, int H, int N, int D, int dtype,
                      int similarity, float* out, void* scratch, size_t scratch_bytes, hipStream_t s,
                      int32_t* status = nullptr);
// the score tail at SD1.5's default tap (256 tokens, head dim 160, 16-bit types): persistent workgroups, K / V streamed once per
// 256 queries through an LDS-DMA ring -- attn160.hip
bool pair_score160_applies(int N, int D, int dtype);
size_t pair_score160_scratch_bytes(int n_pairs, int B, int H);
int launch_pair_score160(const void* q, const void* k, const void* v, const int32_t* idx_a, const int32_t* idx_b, int n_pairs, int B,
                         int H, int mse, float* out, void* scratch, size_t scratch_bytes, hipStream_t s, int32_t* status);
// the same core as a plain SDPA (256 queries = 256 keys, head dim 160, 16-bit types): the U-Net's 16 x 16-level self-attentions
bool sdpa160_applies(const AttnArgs& a);
int launch_sdpa160(const AttnArgs& a, hipStream_t s);

// row-resident fused feed-forward of the 320-channel transformer blocks (h16) -- rowres.hip
//   out = x + W2 (h * gelu(g)) + b2,  [h ; g] = W1 LN(x) + b1
struct FFArgs {
    const void* x = nullptr;                    // [M][C] h16: LayerNorm input and residual
    void* out = nullptr;                        // [M][C] h16 (may alias x)
    const float* ln_g = nullptr;
    const float* ln_b = nullptr;
    const void* stream = nullptr;               // pack_ff_stream output
    const float* b1 = nullptr;                  // [8C] f32, GEGLU-interleaved (pack_vector with geglu_interleave = 1)
    const float* b2 = nullptr;                  // [C] f32
    int M = 0, C = 0;
    float eps = 1e-5f;
    int dtype = DSIM_BF16;                      // DSIM_BF16 or DSIM_F16
};
size_t ff_stream_bytes(int C);                  // 0: no fused kernel for this width
// w1_packed: the GEGLU-interleaved [8C][C] h16 weight (pack_linear with geglu_interleave = 1); w2_packed: [C][4C] h16
int pack_ff_stream(const void* w1_packed, const void* w2_packed, void* stream, int C, hipStream_t s);
int launch_ff_fused(const FFArgs& a, hipStream_t s);

#if !defined(DSIM_H16_IS_F16) && !defined(DSIM_DEVTOOLS)
#define DSIM_HAS_F16_TWINS 1
// the fp16 twins (same sources compiled with -DDSIM_H16_IS_F16); the plain entry points forward compute dtype DSIM_F16 to them
int launch_gemm_f16(const GemmArgs& a, int dtype, hipStream_t s);
int launch_rowlin_f16(const RowLinArgs& a, hipStream_t s);
int launch_ff_fused_f16(const FFArgs& a, hipStream_t s);
int launch_groupnorm_f16(const void* x0, int C0, const void* x1, int C1, const float* gamma, const float* beta, void* out, int B, int HW,
                         int groups, float eps, int silu, int dtype, void* scratch, hipStream_t s);
int launch_groupnorm_pre_f16(const void* x, int C, const float* gamma, const float* beta, void* out, int B, int HW, int groups, float eps,
                             int silu, int dtype, void* scratch, const float* part32, int chunks, hipStream_t s);
int launch_layernorm_f16(const void* x, const float* gamma, const float* beta, void* out, int M, int C, float eps, int dtype, hipStream_t s);
int launch_layernorm_mod_f16(const void* x, const float* scale2, const float* shift2, void* out, int M, int C, int rows_per_batch, float eps,
                             int dtype, hipStream_t s);
int launch_softmax_rows_f16(const void* x, void* out, int rows, int cols, float scale, int dtype, hipStream_t s);
int launch_attention_f16(const AttnArgs& a, int dtype, hipStream_t s);
int launch_pair_score_f16(const void* q, const void* k, const void* v, const int32_t* idx_a, const int32_t* idx_b, int n_pairs, int B, int H,
                          int N, int D, int dtype, int similarity, float* out, void* scratch, size_t scratch_bytes, hipStream_t s,
                          int32_t* status);
#endif

// Per-device once-flags for hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count: the attribute is a
// per-device property of the function, so a process that drives several devices must set it on each.
int cu_count();
struct DeviceOnce {
    unsigned long long done = 0;       // bit d set: attribute applied on device d (d < 64)
    template <typename F> int ensure(F&& apply) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return DSIM_ERR_HIP;
        if (dev >= 0 && dev < 64 && (__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev) & 1ull) return DSIM_OK;
        const int st = apply();          // idempotent: two racing threads may both apply it
        if (st == DSIM_OK && dev >= 0 && dev < 64) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE);
        return st;
    }
};
#define CK_ONCE(once, kern, lds_bytes)                                                                                   \
    do {                                                                                                                 \
        const int _st = (once).ensure([&]() -> int {                                                                     \
            return hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (lds_bytes)) ==  \
                           hipSuccess ? DSIM_OK : DSIM_ERR_HIP;                                                          \
        });                                                                                                              \
        if (_st != DSIM_OK) return _st;                                                                                  \
    } while (0)

}  // namespace dsim
