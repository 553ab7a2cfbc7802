/*
 * diffsim_amd.h -- C ABI of the MI355X-native DiffSim scoring engine (libdiffsim_amd.so).
 *
 * The reference (showlab/DiffSim) has no FFI: its seam is Python.  These entry points are
 * what a binding for the reference's hot path would call; each cites the reference
 * interface it replaces.  Conventions:
 *   - plain pointers and sizes only; every tensor (including the workspace) is allocated
 *     by the caller (PyTorch-ROCm in the shipped wrapper) and passed as a device pointer;
 *   - the callee never allocates after dsim_unet_finalize(), never frees caller memory,
 *     never synchronises the stream and never throws: it returns 0 or a negative
 *     dsim_status code (dsim_strerror() gives the text);
 *   - one handle per device, one host thread per handle, re-entrant across handles (per-kernel launch attributes
 *     are cached per device, so handles on several devices may live in one process);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).
 *
 * Layouts: activations are token-major ("NHWC"): [batch][pixel][channel].  Q/K/V leave the
 * engine as [batch][token][head*head_dim] -- the same memory the reference's non-contiguous
 * (B,H,N,D) views alias (diffsim/hacked_attn.py:74-77).
 */
#ifndef DIFFSIM_AMD_H
#define DIFFSIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSIM_ABI_VERSION 7

typedef enum dsim_status {
    DSIM_OK = 0,
    DSIM_ERR_INVALID = -1,      /* bad argument / unsupported shape */
    DSIM_ERR_MISSING_WEIGHT = -2,
    DSIM_ERR_WORKSPACE = -3,    /* workspace too small */
    DSIM_ERR_HIP = -4,          /* a HIP runtime call failed */
    DSIM_ERR_STATE = -5,        /* call order violated (e.g. qkv before finalize) */
    DSIM_ERR_NO_DEVICE = -6
} dsim_status;

typedef enum dsim_dtype { DSIM_F32 = 0, DSIM_BF16 = 1, DSIM_F16 = 2 } dsim_dtype;

/* where the hooked attn1 sits: diffsim/diffsim.py:122-145
 *   DOWN: unet.down_blocks[:-1][layer]   MID: unet.mid_block   UP: unet.up_blocks[1:][layer]
 * always ...attentions[-1].transformer_blocks[-1].attn1                                   */
typedef enum dsim_tap_block { DSIM_TAP_DOWN = 0, DSIM_TAP_MID = 1, DSIM_TAP_UP = 2 } dsim_tap_block;

#define DSIM_MAX_LEVELS 4

/* Subset of diffusers' unet/config.json the path depends on. */
typedef struct dsim_unet_cfg {
    int32_t in_channels;                          /* 4 */
    int32_t n_levels;                             /* 4 */
    int32_t block_out_channels[DSIM_MAX_LEVELS];  /* 320,640,1280,1280 */
    int32_t down_has_attn[DSIM_MAX_LEVELS];       /* 1,1,1,0  (CrossAttnDownBlock2D vs DownBlock2D) */
    int32_t up_has_attn[DSIM_MAX_LEVELS];         /* 0,1,1,1 */
    int32_t layers_per_block;                     /* 2 */
    int32_t num_heads;                            /* 8 */
    int32_t cross_attention_dim;                  /* 768 */
    int32_t norm_num_groups;                      /* 32 */
    float   norm_eps;                             /* 1e-5 (ResnetBlock2D); Transformer2D GN uses 1e-6 */
    int32_t sample_size;                          /* latent side: 64 for 512 px */
    int32_t ctx_len;                              /* 77 */
    int32_t compute_dtype;                        /* dsim_dtype: DSIM_F32 (parity mode) or DSIM_BF16 */
    int32_t tap_block;                            /* dsim_tap_block */
    int32_t tap_layer;                            /* ABSOLUTE index into down_blocks / up_blocks (the wrapper
                                                     resolves the reference's slices: SD1.5 down[:-1]/up[1:],
                                                     SDXL down[1:]/up[:-1]); ignored for DSIM_TAP_MID */
    int32_t tap_attn;                             /* attention index inside the block, -1 = last (SD1.5) */
    int32_t tap_tfm;                              /* transformer_block index inside it, -1 = last (SD1.5) */
    /* SDXL deltas; zero = SD1.5 behaviour */
    int32_t heads_per_level[DSIM_MAX_LEVELS];     /* 5,10,20 ; 0 = num_heads everywhere */
    int32_t depth_per_level[DSIM_MAX_LEVELS];     /* transformer blocks per Transformer2DModel: 1,2,10 ; 0 = 1 */
    int32_t addition_embed;                       /* 1: "text_time" added conditioning (add_embedding.*) */
    int32_t addition_time_embed_dim;              /* 256 */
    int32_t pooled_dim;                           /* 1280 */
} dsim_unet_cfg;

typedef struct dsim_unet dsim_unet;

int         dsim_version(void);
const char* dsim_strerror(int status);
/* number of visible HIP devices (does not initialise a context beyond the count query) */
int         dsim_device_count(void);

/* ---- U-Net-to-tap engine: replaces DiffSimPipeline.step()'s `self.unet(...)` call
 *      (diffsim/diffsim_pipeline.py:213-221) plus the pre-hook that stashes q,k,v
 *      (diffsim/diffsim.py:43-56 -> diffsim/hacked_attn.py:61-77). ---------------------- */
int  dsim_unet_create(const dsim_unet_cfg* cfg, dsim_unet** out);
void dsim_unet_destroy(dsim_unet* h);

/* Hand one parameter over under its diffusers state-dict key (e.g.
 * "down_blocks.0.resnets.0.conv1.weight").  `dev_ptr` is borrowed until dsim_unet_finalize()
 * returns; src dtype may be f32, bf16 or f16; shape is the diffusers shape.  Parameters
 * that lie after the tap are accepted and ignored. */
int  dsim_unet_load_weight(dsim_unet* h, const char* key, const void* dev_ptr, int dtype,
                           const int64_t* shape, int ndim);
/* Repack every parameter up to the tap into the engine's own device buffers (conv weights
 * [Cout][3][3][Cin], fused QKV / KV, GEGLU-interleaved FF).  Allocates; synchronises `stream`. */
int  dsim_unet_finalize(dsim_unet* h, void* stream);
/* Diffusion timestep t (an actual timestep, not the reference's --target_step index; the
 * wrapper maps index -> t through the PNDM table, diffsim/diffsim_pipeline.py:153-157).
 * Pre-computes the time embedding and every ResnetBlock2D time_emb_proj (t is constant over a
 * run).  Enqueues on `stream`. */
int  dsim_unet_set_timestep(dsim_unet* h, int t, void* stream);
/* SDXL: timestep plus the added conditioning of DiffSimXLPipeline.step (diffsim/diffsim_xl_pipeline.py:231-262,
 * 312): text_embeds f32 device [2][pooled_dim] = [negative, positive] pooled prompt embeddings, time_ids f32
 * device [2][6] = (original_size, crop_top_left, target_size).  The two CFG halves get different time
 * embeddings; every ResnetBlock2D bias is prepared once per half. */
int  dsim_unet_set_conditioning(dsim_unet* h, int t, const float* text_embeds, const float* time_ids, void* stream);

/* Bytes of workspace one dsim_unet_qkv call over n_images needs; 0 when the call is impossible: handle not
 * finalized, or some activation of that batch would reach 2 GiB (tensors are addressed with 32-bit offsets) --
 * split the batch then. */
size_t dsim_unet_workspace_bytes(const dsim_unet* h, int n_images);

/* One noised U-Net forward to the tap for n_images latents, each duplicated for
 * classifier-free guidance ([uncond, cond], diffsim/diffsim_pipeline.py:208):
 *   latents, noise : f32 [n_images][C_in][s][s]  (NCHW, as the reference holds them)
 *   x_t = sqrt_abar*latents + sqrt_1m_abar*noise  (scheduler.add_noise, pipeline :177-183)
 *   ctx            : f32 [2][ctx_len][cross_attention_dim] = [uncond, cond] prompt embeddings
 *   q,k,v (out)    : compute_dtype [n_images][2][tokens][heads*head_dim]
 */
int  dsim_unet_qkv(dsim_unet* h, const float* latents, const float* noise, float sqrt_abar,
                   float sqrt_1m_abar, const float* ctx, int n_images, void* q, void* k, void* v,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Measurement aid (bench.py's roofline leg): while enabled, dsim_unet_qkv brackets every kernel
 * launch with a pair of HIP events recorded on the launch stream.  After the caller has
 * synchronised the stream, dsim_unet_profile_get returns, per launch: the kernel family name
 * (e.g. "gemm_bf16_256x160_conv3"), its ALGORITHMIC flops and bytes, and the elapsed ms.
 * dsim_unet_profile(h, enable) clears earlier records.  Not for use inside a timed region. */
int  dsim_unet_profile(dsim_unet* h, int enable);
int  dsim_unet_profile_count(const dsim_unet* h);
int  dsim_unet_profile_get(dsim_unet* h, int i, char* name, int name_cap, double* flops,
                           double* bytes, double* ms);

/* geometry of the tap for the current cfg: tokens, heads, head_dim */
int  dsim_unet_tap_shape(const dsim_unet* h, int* tokens, int* heads, int* head_dim);

/* Move the tap of a finalized handle (same meaning as the cfg fields of the same names): the packed weights are
 * shared by every tap, only the point where the walk stops changes -- one weight copy serves
 * --target_block/--target_layer sweeps (diffsim/diffsim.py:122-145, diffsim/diffsim_xl.py:88-107).
 * DSIM_ERR_MISSING_WEIGHT when a parameter needed before the new tap was never loaded (the old tap stays). */
int  dsim_unet_set_tap(dsim_unet* h, int tap_block, int tap_layer, int tap_attn, int tap_tfm);
/* Opt-in: compute what the two classifier-free-guidance halves share once.  The reference runs torch.cat([latents] * 2)
 * through the whole U-Net (diffsim_pipeline.py:208-221); conv_in, the first ResnetBlock2D and the first transformer up to its
 * cross-attention query see two bit-identical halves (one time embedding: SD1.5 graphs only; ignored for SDXL and when the tap
 * lies in the first down block).  Scores are bit-identical to the default; 6 % fewer FLOPs are executed. */
int  dsim_unet_set_cfg_dedup(dsim_unet* h, int enable);
/* Which multi-operator kernels replace their unfused chains (bf16 handles; default DSIM_FUSE_ALL).  0 runs every layer as its own
 * launch -- the A/B switch of bench.py --fusion and of the parity tests; results agree to bf16 rounding, not bit for bit.
 * DSIM_FUSE_FF: norm3 -> ff.net.0.proj (GEGLU) -> ff.net.2 -> + residual of a 320-channel BasicTransformerBlock as one launch
 * (hacked_modules.py:118-132).
 * DSIM_FUSE_LNPROJ: the LayerNorm in front of a 320-channel block's self-attention q/k/v projection (norm1 -> to_q|to_k|to_v)
 * and of its cross-attention query (norm2 -> attn2.to_q) runs inside the projection's launch (hacked_modules.py:88-116).
 * DSIM_FUSE_TAPQKV: the tapped layer's to_q / to_k / to_v (hacked_attn.py:61-69) as ONE N = 3C launch whose column runs go to the
 * three output tensors -- taken when q, k, v lie at equal distances in memory (one [3][...] allocation); bit-identical to the
 * three launches, any compute dtype. */
#define DSIM_FUSE_FF     1
#define DSIM_FUSE_LNPROJ 2
#define DSIM_FUSE_TAPQKV 4
#define DSIM_FUSE_ALL    7
int  dsim_unet_set_fusion(dsim_unet* h, int mask);
/* Latent side of the next dsim_unet_qkv calls (cfg.sample_size is only the default): the reference runs any
 * --image_size through the same weights (argprocess.py:8: default 512 px, SDXL native 1024 px).  `side` must be a
 * multiple of 2^(n_levels-1). */
int  dsim_unet_set_sample_size(dsim_unet* h, int side);

/* ---- score tail: replaces diffsim/diffsim.py:177-197 (4x F.scaled_dot_product_attention,
 *      2x F.cosine_similarity or F.mse_loss, mean).  Fused: the O tensors never reach HBM. --
 *   q,k,v       : dtype [n_feat][B][N][H*D]   (features of n_feat images, B = CFG batch = 2)
 *   idx_a,idx_b : device int32 [n_pairs]; pair p scores image idx_a[p] against idx_b[p]
 *   similarity  : 0 cosine, 1 mse
 *   out_scores  : device f32 [n_pairs]
 */
size_t dsim_pair_score_workspace_bytes(int n_pairs, int B, int H, int N, int D);
int    dsim_pair_score(const void* q, const void* k, const void* v, const int32_t* idx_a,
                       const int32_t* idx_b, int n_pairs, int B, int H, int N, int D, int dtype,
                       int similarity, float* out_scores, void* workspace, size_t workspace_bytes,
                       void* stream);
/* The same call with a per-pair status: status[p] = 0 when score p is finite, 1 when it is NaN or infinite
 * (non-finite features: an overflowed activation or a corrupt weight; the reference would print the NaN and
 * count the triplet as wrong, cute_main.py:201-205).  status: device int32 [n_pairs]. */
int    dsim_pair_score_status(const void* q, const void* k, const void* v, const int32_t* idx_a,
                              const int32_t* idx_b, int n_pairs, int B, int H, int N, int D, int dtype,
                              int similarity, float* out_scores, int32_t* status, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- VAE encoder: replaces `pipe.vae.encode(image)` in
 *      DiffSim.prepare_image_latents (diffsim/diffsim.py:92-96).  Sampling
 *      z = mean + exp(0.5*clamp(logvar,-30,20))*eps and the 0.18215 scaling stay with the caller,
 *      which owns the generator whose draw order defines the score. ------------------------------ */
typedef struct dsim_vae_cfg {
    int32_t in_channels;                          /* 3 */
    int32_t latent_channels;                      /* 4 */
    int32_t n_levels;                             /* 4 */
    int32_t block_out_channels[DSIM_MAX_LEVELS];  /* 128,256,512,512 */
    int32_t layers_per_block;                     /* 2 */
    int32_t norm_num_groups;                      /* 32 */
    int32_t compute_dtype;                        /* DSIM_F32 or DSIM_BF16 */
} dsim_vae_cfg;
typedef struct dsim_vae dsim_vae;

int    dsim_vae_create(const dsim_vae_cfg* cfg, dsim_vae** out);
void   dsim_vae_destroy(dsim_vae* h);
/* diffusers AutoencoderKL keys: "encoder.*" and "quant_conv.*" (decoder keys are not needed) */
int    dsim_vae_load_weight(dsim_vae* h, const char* key, const void* dev_ptr, int dtype,
                            const int64_t* shape, int ndim);
int    dsim_vae_finalize(dsim_vae* h, void* stream);
size_t dsim_vae_workspace_bytes(const dsim_vae* h, int n_images, int image_size);
/* images: f32 [n][3][S][S] in [-1,1] (process_image output); moments (out): f32 [n][2*latent][S/8][S/8]
 * = cat(mean, logvar) exactly as AutoencoderKL's quant_conv output */
int    dsim_vae_encode(dsim_vae* h, const float* images, int n_images, int image_size, float* moments,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Measurement aid (ABI v7), same contract and record format as dsim_unet_profile / _count / _get: per launch of the next
 * dsim_vae_encode calls, the kernel family, its algorithmic FLOPs / bytes and its HIP-event duration on the launch stream */
int    dsim_vae_profile(dsim_vae* h, int enable);
int    dsim_vae_profile_count(const dsim_vae* h);
int    dsim_vae_profile_get(dsim_vae* h, int i, char* name, int name_cap, double* flops, double* bytes, double* ms);

/* ---- DiT-XL/2 scorer backbone: replaces `diffusion.p_sample(model, latents, t,
 *      model_kwargs=dict(y=[1, 1000]))` + the pre-hook on model.blocks[L].attn of diffsim/diffsim_dit.py:93-114.
 *      Weights under the reference's own state-dict keys (DiT/modelsdit.py): pos_embed, x_embedder.proj.*,
 *      t_embedder.mlp.{0,2}.*, y_embedder.embedding_table.weight, blocks.N.{attn.qkv,attn.proj,mlp.fc1,mlp.fc2,
 *      adaLN_modulation.1}.* -------------------------------------------------------------------------------- */
typedef struct dsim_dit_cfg {
    int32_t input_size;        /* latent side: 32 for 256 px */
    int32_t patch_size;        /* 2 */
    int32_t in_channels;       /* 4 */
    int32_t hidden_size;       /* 1152 */
    int32_t depth;             /* 28 */
    int32_t num_heads;         /* 16 */
    int32_t mlp_ratio;         /* 4 */
    int32_t num_classes;       /* 1000 (embedding table has num_classes + 1 rows) */
    int32_t freq_dim;          /* 256 */
    int32_t compute_dtype;     /* DSIM_F32 or DSIM_BF16 */
    int32_t tap_layer;         /* block index L of the hooked attention */
} dsim_dit_cfg;
typedef struct dsim_dit dsim_dit;

int    dsim_dit_create(const dsim_dit_cfg* cfg, dsim_dit** out);
void   dsim_dit_destroy(dsim_dit* h);
int    dsim_dit_load_weight(dsim_dit* h, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim);
int    dsim_dit_finalize(dsim_dit* h, void* stream);
/* t_model = the timestep the MODEL sees = SpacedDiffusion.timestep_map[1000 - target_step]
 * (DiT/diffusion/respace.py:117-129); y0,y1 = class labels of the two batch halves (1 and num_classes = null) */
int    dsim_dit_set_conditioning(dsim_dit* h, int t_model, int y0, int y1, void* stream);
/* Attention arithmetic of the DiT blocks: 0 = the handle's compute dtype (default), 1 = fp8 (OCP e4m3) MFMAs for
 * both QK^T and PV with f32 softmax (BASELINE.json config 5; bf16 handles, head_dim 72 or 32 only).  The
 * reference's timm Attention (DiT/modelsdit.py:103-124, F.scaled_dot_product_attention in fp16) has no such mode:
 * it is an opt-in accuracy/throughput trade, compared with the oracle under a stated looser tolerance. */
int    dsim_dit_set_attention(dsim_dit* h, int mode);
/* Measurement aid, same contract and record format as dsim_unet_profile / _count / _get */
int    dsim_dit_profile(dsim_dit* h, int enable);
int    dsim_dit_profile_count(const dsim_dit* h);
int    dsim_dit_profile_get(dsim_dit* h, int i, char* name, int name_cap, double* flops, double* bytes, double* ms);
size_t dsim_dit_workspace_bytes(const dsim_dit* h, int n_images);
/* Move the tap (the block whose attention q,k,v are emitted) without re-packing: ONE weight copy serves every
 * --target_layer of diffsim_DiT.diffsim_score (diffsim/diffsim_dit.py:100-104 hooks model.blocks[target_layer[0]].attn).
 * DSIM_ERR_MISSING_WEIGHT if a block up to the new tap was never loaded (the old tap stays).  dsim_dit_set_conditioning
 * prepares the modulation vectors of every loaded block, so the conditioning survives a move of the tap. */
int    dsim_dit_set_tap(dsim_dit* h, int tap_layer);
/* x_t = sqrt_abar*latents + sqrt_1m_abar*noise (DDIM add_noise at t = target_step, diffsim_dit.py:63-72);
 * q,k,v (out): compute dtype [n_images][2][tokens][heads*head_dim] */
int    dsim_dit_qkv(dsim_dit* h, const float* latents, const float* noise, float sqrt_abar, float sqrt_1m_abar,
                    int n_images, void* q, void* k, void* v, void* workspace, size_t workspace_bytes, void* stream);

/* ---- the arithmetic either side of the VAE encoder, on the device (the host only decodes and resizes) ----
 * dsim_image_preprocess: what process_image does after its Lanczos resize (diffsim/diffsim.py:31-41): pixels u8 [n][H][W][3] ->
 *   f32 [n][3][H][W] = (p / 255 - 0.5) / 0.5 in IEEE f32, bit-identical to the numpy path; to_half = 1 additionally rounds
 *   through fp16 (the SD1.5 pipeline's `image.to(dtype=float16)`, diffsim.py:93).
 * dsim_latent_sample: `scaling_factor * latent_dist.sample()` of prepare_image_latents (diffsim.py:92-96): out[j] =
 *   sf * (mean + exp(0.5 clamp(logvar, -30, 20)) * eps) for image first + j * stride of `moments` ([n][2C][hw], dsim_vae_encode's
 *   output); eps f32 [eps_n][C][hw] drawn by the CALLER's generator (eps_n = 1: one draw shared by all, as every reference call
 *   reseeds; or one per output image); round_fp16 = 1 rounds the latents through fp16 (diffsim_xl.py:63, the fp16 pipelines). */
int dsim_image_preprocess(const unsigned char* pixels_hwc, float* out, int n, int H, int W, int to_half, void* stream);
int dsim_latent_sample(const float* moments, const float* eps, float* out, int n_out, int first, int stride, int C, int hw,
                       int eps_n, float scaling_factor, int round_fp16, void* stream);

/* ---- single-operator entry points (kernel-level parity tests and micro-benchmarks) -----
 * x: dtype [M][K] (or NHWC image for the conv forms); w: diffusers-layout f32 weight.      */
int dsim_op_linear(const void* x, const float* w /*[N][K]*/, const float* bias /*[N] or NULL*/,
                   const void* residual /*[M][N] or NULL*/, void* out /*[M][N]*/, int M, int N,
                   int K, int dtype, int geglu /* w is [2*N][K], out = h*gelu(g) */, void* stream);
int dsim_op_conv3x3(const void* x /*[B][H][W][Cin]*/, const float* w /*[Cout][Cin][3][3]*/,
                    const float* bias, const void* residual, void* out, int B, int H, int W,
                    int Cin, int Cout, int stride, int upsample, int dtype, void* stream);
int dsim_op_groupnorm(const void* x0, int C0, const void* x1, int C1, const float* gamma,
                      const float* beta, void* out, int B, int HW, int groups, float eps,
                      int silu, int dtype, void* stream);
int dsim_op_layernorm(const void* x, const float* gamma, const float* beta, void* out, int M,
                      int C, float eps, int dtype, void* stream);
/* q: [B][Nq][ldq] at column offset h*D; k,v: [Bkv][Nk][ldk]; batch b reads kv batch b % Bkv */
int dsim_op_attention(const void* q, int ldq, const void* k, const void* v, int ldk, void* out,
                      int ldo, int B, int Bkv, int H, int Nq, int Nk, int D, int dtype,
                      void* stream);
/* the same attention on bf16 tensors with fp8 (e4m3) MFMAs -- the kernel behind dsim_dit_set_attention(h, 1); D = 72 or 32 */
int dsim_op_attention_fp8(const void* q, int ldq, const void* k, const void* v, int ldk, void* out,
                          int ldo, int B, int Bkv, int H, int Nq, int Nk, int D, void* stream);

/* One BasicTransformerBlock feed-forward as a single launch (bf16, C = 320): out = x + ff.net.2(GEGLU(ff.net.0.proj(LayerNorm(x))))
 * -- the chain /root/reference/diffsim/hacked_modules.py:118-132 runs as norm3 -> ff -> + hidden_states.  w1: [8C][C] f32 (diffusers
 * ff.net.0.proj.weight, rows [h ; g]), b1: [8C], w2: [C][4C], b2: [C]; x / out: bf16 [M][C] (out may alias x).
 * Returns DSIM_ERR_INVALID for a width the fused kernel does not cover.                                                   */
int dsim_op_ff_fused(const void* x, const float* ln_gamma, const float* ln_beta, const float* w1, const float* b1,
                     const float* w2, const float* b2, void* out, int M, int C, float eps, void* stream);
/* LayerNorm + bias-free Linear as a single launch (bf16, C = 320, N a multiple of 64 up to 960): out[M][N] = LayerNorm(x) W^T --
 * norm1 -> to_q|to_k|to_v and norm2 -> attn2.to_q of /root/reference/diffsim/hacked_modules.py:88-116.  w: [N][C] f32;
 * ln_gamma = ln_beta = NULL skips the LayerNorm.  Returns DSIM_ERR_INVALID for a shape the kernel does not cover.            */
int dsim_op_ln_linear(const void* x, const float* ln_gamma, const float* ln_beta, const float* w, void* out, int M, int C,
                      int N, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSIM_AMD_H */
