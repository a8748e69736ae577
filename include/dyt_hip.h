/*
 * dyt_hip.h -- C ABI of libdyt_hip.so: the MI355X (gfx950) implementation of the
 * Dynamic-Tuning ViT-B/16 fine-tune hot path.
 *
 * The reference (NUS-HPC-AI-Lab/Dynamic-Tuning) has no FFI: its hot path sits behind a
 * Python nn.Module API.  This header is therefore the boundary a
 * maintainer binds with ctypes from the reference's module files; every entry point cites
 * the reference interface it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 (DYT_OK) or a negative error code; no exceptions cross the
 *     ABI; dyt_last_error() returns a thread-local message for the last failure;
 *   - all tensor arguments are DEVICE pointers owned by the caller (torch allocates them);
 *     the library never frees them and never synchronises the host with the device;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - the library owns (hipMalloc) only its private copies of the frozen weights and its
 *     activation workspace, both sized at dyt_ctx_create();
 *   - one context per process / device; thread-compatible, not thread-safe.
 */
#ifndef DYT_HIP_H
#define DYT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYT_OK 0
#define DYT_ERR_ARG (-1)     /* bad argument / unsupported shape */
#define DYT_ERR_HIP (-2)     /* a HIP runtime call failed */
#define DYT_ERR_STATE (-3)   /* call order violated (e.g. backward without a saved forward) */

/* arithmetic mode of the dense contractions (dyt_config::precision).  The library is built twice from the same sources:
 * libdyt_hip.so (16-bit operand type bfloat16) and libdyt_hip_f16.so (-DDYT_FP16: IEEE half; dyt_operand_type() tells which).
 * Together with DYT_OPT_F32_SPLIT16 (below) that gives the eight host-side precision names of dynamic-tuning_amd/runtime.py:
 *   "fp32"     either library, DYT_PREC_FP32                      exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32)
 *   "bf16"     libdyt_hip.so,     DYT_PREC_BF16                   bf16 MFMA operands, fp32 accumulate / residual stream / statistics
 *   "fp16"     libdyt_hip_f16.so, DYT_PREC_BF16 (its 16-bit mode) IEEE-half operands, gradient operands x 2^12 (the default)
 *   "fp16x3"   libdyt_hip_f16.so, DYT_PREC_FP32 + SPLIT16 = 1     every frozen-weight product as three half products, fp32 data flow
 *   "fp16x3f"  ... SPLIT16 = 2                                    the same forward, gradient products hi * hi (fp32-width backward)
 *   "fp16x3h"  ... SPLIT16 = 3                                    the same forward, backward pass on 16-bit operands (fp16 mode's kernels)
 *   "fp16x3q"  ... SPLIT16 = 5                                    as 3 with qkv / proj as hi * hi + two fp8 correction products
 *   "fp16f8"   ... SPLIT16 = 4                                    as 3 with every forward GEMM in that fp8-correction form
 * In the two 16-bit modes LayerNorm-2 is folded into the fc1 GEMM (per-row statistics from the proj epilogue, gamma / beta folded into
 * the frozen fc1 weight copy, normalisation in the fc1 epilogue: DESIGN.md section 5).  DYT_LN_FOLD=0 in the environment when
 * dyt_create runs keeps it a separate kernel (LayerNorm in fp32, then rounded: the order of the reference's autocast). */
#define DYT_PREC_FP32 0 /* exact: fp32 operands, fp32 accumulate -- the parity mode */
#define DYT_PREC_BF16 1 /* fast: 16-bit MFMA operands (bfloat16 or IEEE half by build), fp32 accumulate, fp32 residual stream */

/* dyt_forward flags */
#define DYT_F_TRAINING 1       /* model.train(): Gumbel noise + adapter dropout (dynamic_adapter.py:29-42,127) */
#define DYT_F_COMPLETE 2       /* forward(x, complete_model=True): mask not applied (vision_transformer_IN21K.py:161) */
#define DYT_F_SAVE 4           /* keep activations of this pass in `slot` for dyt_backward */
#define DYT_F_MASKED_DENSE 8   /* student pass computes the MLP for every token and multiplies by the
                                  mask, exactly as the reference trains (vision_transformer_IN21K.py:159-162), so
                                  the gate of a DROPPED token still receives <dL/dx', mlp(x)>; its backward is
                                  compacted all the same (rows of mask * dL/dx' of dropped tokens are exactly zero):
                                  the reference's gradients at 133.5 instead of 139.6 GFLOP / image.
                                  Default is the compacted MLP of models/model_speed_test.py:274-310, forward too */
#define DYT_F_ACCUM_GRAD 32    /* dyt_step_fwd_bwd: add to grad_flat instead of overwriting it -- gradient accumulation over
                                  accum_iter micro-batches (engine_finetune.py:43-46,66-76; fold 1/accum_iter into dyt_adamw's grad_scale) */
#define DYT_F_DEVICE_SEED 64   /* the Philox seed is read from the context's device-side seed word (dyt_seed) instead of the
                                  `seed` argument, and dyt_step_fwd_bwd advances that word by one when it is done: a step
                                  captured into a hipGraph then draws fresh Gumbel noise / dropout masks at every replay */
#define DYT_F_TOKENS_IN 128    /* dyt_forward: `images` is the residual stream [B,197,768] itself (no patch embedding) -- stand-alone
                                * Block.forward on a token tensor, as block_flops_dict.py:36-46 calls it; forward only */
#define DYT_F_TOKENS_OUT 256   /* dyt_forward: `logits` receives the block stack's output tokens [B,197,768] (no final norm / head) */
#define DYT_F_GATE_ALWAYS 16   /* also evaluate the token dispatcher in a COMPLETE pass (the reference does,
                                  and discards it, :150-152) so token_select/token_logits are returned */

/* parameter ids (frozen unless marked T = trainable); `layer` is the block index, or 0 */
enum dyt_param {
    DYT_P_CLS = 0,      /* cls_token [1,1,768] */
    DYT_P_POS,          /* pos_embed [1,197,768] */
    DYT_P_PE_W,         /* patch_embed.proj.weight [768,3,16,16] */
    DYT_P_PE_B,         /* patch_embed.proj.bias [768] */
    DYT_P_LN1_W, DYT_P_LN1_B,     /* blocks.i.norm1 */
    DYT_P_QKV_W, DYT_P_QKV_B,     /* blocks.i.attn.qkv [2304,768] */
    DYT_P_PROJ_W, DYT_P_PROJ_B,   /* blocks.i.attn.proj [768,768] */
    DYT_P_LN2_W, DYT_P_LN2_B,     /* blocks.i.norm2 */
    DYT_P_FC1_W, DYT_P_FC1_B,     /* blocks.i.mlp.fc1 [3072,768] */
    DYT_P_FC2_W, DYT_P_FC2_B,     /* blocks.i.mlp.fc2 [768,3072] */
    DYT_P_NORM_W, DYT_P_NORM_B,   /* norm */
    DYT_P_AD_DOWN_W, DYT_P_AD_DOWN_B, /* T blocks.i.adaptmlp.down_proj [r,768] */
    DYT_P_AD_UP_W, DYT_P_AD_UP_B,     /* T blocks.i.adaptmlp.up_proj [768,r] */
    DYT_P_GATE_W, DYT_P_GATE_B,       /* T blocks.i.mlp_token_select.mlp_head [1,768] */
    DYT_P_HEAD_W, DYT_P_HEAD_B,       /* T head [C,768] */
    /* video model only (cfg.frames > 1): the attentive pooling head, all trainable (missing from the
     * ViT checkpoint) -- video_models/video_vision_transformer_IN21K.py:27-110,407-410 */
    DYT_P_POOL_QUERY,                     /* T query_token [1,1,768] */
    DYT_P_POOL_NQ_W, DYT_P_POOL_NQ_B,     /* T attentive_blocks.norm_q */
    DYT_P_POOL_NK_W, DYT_P_POOL_NK_B,     /* T attentive_blocks.norm_k */
    DYT_P_POOL_NV_W, DYT_P_POOL_NV_B,     /* T attentive_blocks.norm_v */
    DYT_P_POOL_Q_W, DYT_P_POOL_K_W, DYT_P_POOL_V_W,   /* T attentive_blocks.cross_attn.{q,k,v}.weight [768,768] */
    DYT_P_POOL_Q_BIAS, DYT_P_POOL_V_BIAS,             /* T attentive_blocks.cross_attn.{q_bias,v_bias} [768] */
    DYT_P_POOL_PROJ_W, DYT_P_POOL_PROJ_B,             /* T attentive_blocks.cross_attn.proj [768,768] */
    DYT_P_AD_SCALE,     /* T blocks.i.adaptmlp.scale [1] -- only with tuning_config.ffn_adapter_scalar == "learnable_scalar" (DYT_OPT_LEARNABLE_SCALE) */
    DYT_P_AD_LN_W, DYT_P_AD_LN_B,   /* T blocks.i.adaptmlp.adapter_layer_norm_before.{weight,bias} [768] -- only with dyt_config::adapter_ln != 0 */
    DYT_P_COUNT
};

/* Replaces the constructor arguments of vit_base_patch16_224_in21k(num_classes, tuning_config,
 * select_config) -- models/vision_transformer_IN21K.py:199-323,414-421.  Fixed by the factory:
 * patch 16, dim 768, depth 12, heads 12, mlp_ratio 4, qkv_bias, LayerNorm eps 1e-6. */
typedef struct dyt_config {
    int32_t num_classes;     /* head width C */
    int32_t ffn_num;         /* adapter bottleneck r (tuning_config.ffn_num), 1..64 */
    int32_t depth;           /* 12 */
    int32_t precision;       /* DYT_PREC_* */
    int32_t max_batch;       /* images per call the workspace is sized for */
    int32_t slots;           /* saved-activation slots (2: student + teacher pass) */
    float adapter_scale;     /* tuning_config.ffn_adapter_scalar (0.1 main_image.py:192, 1.0 main_vtab.py:185) */
    float adapter_dropout;   /* 0.1, vision_transformer_IN21K.py:133 */
    float tau;               /* 5, dynamic_adapter.py:59 */
    float threshold;         /* 0.5, dynamic_adapter.py:59 */
    int32_t frames;          /* 0/1: image model.  t > 1: video model (video_models/video_vision_transformer_IN21K.py:
                                435-483) -- `images`/`batch` of every call are the b*t frames of b clips, clip-major
                                ("b c t h w -> (b t) c h w"), batch % t == 0; logits / targets have b = batch/t rows;
                                the cls-pooling head is replaced by the attentive pooling head over the t*197
                                final-norm tokens of a clip (one query, 12 heads) */
    int32_t adapter_ln;      /* tuning_config.ffn_adapter_layernorm_option (models/dynamic_adapter.py:88,95-98,121-122,132-133): 0 = "none" (every
                                shipped script), 1 = "in": the adapter reads LayerNorm(u) (its own trainable nn.LayerNorm(768), eps 1e-5), 2 = "out":
                                the scaled adapter output goes through that LayerNorm before it joins the residual stream.  Adds the two
                                [768] tensors DYT_P_AD_LN_W / _B per block to the flat trainable layout.  Generic row kernels (no fusion, no
                                cls-only tail); image model only.  (ABI v2: the field was appended in round 6.) */
} dyt_config;

typedef struct dyt_ctx dyt_ctx;

const char* dyt_last_error(void);
int dyt_version(void);
/* the 16-bit operand type DYT_PREC_BF16 selects in THIS build of the library: 0 bfloat16 (libdyt_hip.so), 1 IEEE half
 * (libdyt_hip_f16.so, same sources compiled with -DDYT_FP16; the reference's own autocast dtype, engine_finetune.py:47) */
int dyt_operand_type(void);

int dyt_ctx_create(const dyt_config* cfg, dyt_ctx** out);
int dyt_ctx_destroy(dyt_ctx* ctx);
/* bytes of device memory the context holds (weights + workspace) */
int dyt_ctx_bytes(const dyt_ctx* ctx, int64_t* bytes);

/* Scheduling options (results are identical either way; both default to on):
 *   DYT_OPT_STREAM_OVERLAP  0: everything on the caller's stream.  1 (default): dyt_step_fwd_bwd runs the student and the
 *                           teacher pass on two streams (fork/join with events on the caller's stream).  2: additionally
 *                           the adapter branch of every block on its own stream, 3: only that -- both measured SLOWER
 *                           than 1 whenever the branch streams really get their own hardware queues (26.1 vs 31.5 /
 *                           30.0 ms per step at B=128 with GPU_MAX_HW_QUEUES=8; the serial step is 29.5 ms), kept for
 *                           measurement only.  4: the two FORWARD passes overlap, the teacher's backward starts after the
 *                           student's -- like 0 a bit-reproducible schedule, 29.0 instead of 26.1 ms
 *   DYT_OPT_CLS_TAIL        last block: evaluate MLP + adapter (forward and backward) for the cls rows
 *                           only -- only x[:,0] reaches forward_head (vision_transformer_IN21K.py:375-380)
 *   DYT_OPT_SHARE_BLOCK0    dyt_step_fwd_bwd: the teacher pass reuses the student pass's patch embedding and
 *                           block-0 attention branch (same images, frozen weights, nothing random before it) */
#define DYT_OPT_STREAM_OVERLAP 1
#define DYT_OPT_CLS_TAIL 2
#define DYT_OPT_SHARE_BLOCK0 3
/*   DYT_OPT_COUNT_FLOPS_TOKENS  value n in 1..197 (0 = off): the FLOP-probe variant of the reference,
 *                           Block.forward_count_flops (vision_transformer_IN21K.py:167-185, driven by
 *                           block_flops_dict.get_block_flops :33-55 through `count_flops` / `token_select_num`): every
 *                           block runs its MLP on the FIRST n tokens of each image (cls + the first n-1 patch tokens)
 *                           whatever the gate decides; token_select reports that forced pattern.  This one DOES change
 *                           results. */
#define DYT_OPT_COUNT_FLOPS_TOKENS 4
/*   DYT_OPT_GRAD_SCALE_LOG2     value k in 0..24: the gradient is multiplied by 2^k wherever the library holds it in its 16-bit
 *                               operand type (g_at, dZ, dA2, dqkv ...) and divided again where it returns to fp32 -- a fixed loss
 *                               scale that never leaves the library (what the reference's GradScaler does for its fp16 autocast,
 *                               misc.py:252-272).  Default: 0 in libdyt_hip.so (bf16 has fp32's exponent range), 12 in
 *                               libdyt_hip_f16.so.  Returned gradients are unscaled either way. */
#define DYT_OPT_GRAD_SCALE_LOG2 5
/*   DYT_OPT_FC2_CAT             1 (default): in the 16-bit modes the adapter's up-projection is computed as the leading
 *                           k-tile of the fc2 contraction (x_out = u + [d_act | h][s Wup | W2]^T + b: one pass over the
 *                           fp32 residual stream instead of two, no up-projection launch) in every compacted or
 *                           complete pass (training student pass: the gate gradient is corrected for the adapter term).  0: always two launches (reference op
 *                           order, models/vision_transformer_IN21K.py:157-163).  fp32 mode: ignored (always two launches). */
#define DYT_OPT_FC2_CAT 6
/*   DYT_OPT_ATTN_BWD_FUSED      1 (default): 16-bit modes run the attention backward of a head (dQ and dK/dV) in ONE persistent
 *                           kernel (q, dO, o read once; delta never leaves the chip); 0: two kernels (bit-identical results);
 *                           2: the fused kernel with the per-wave k / v / o rows prefetched a head ahead.  PROCESS-wide. */
#define DYT_OPT_ATTN_BWD_FUSED 7
/*   DYT_OPT_F32_SPLIT16         fp32 mode only, default 0.  1: every GEMM against a FROZEN weight runs on the 16-bit matrix cores as
 *                           three products hi*hi + hi*lo + lo*hi (both fp32 operands split into two 16-bit parts, one contraction
 *                           over the K-concatenated parts, fp32 accumulate, the fp32 epilogues): the per-GEMM error of the exact
 *                           fp32 MFMA kernel (1-2e-6 of max|C| with IEEE-half parts) at ~2.7x its speed; the attention forward and
 *                           backward likewise (DYT_SPLIT_ATTN=0 keeps the exact-fp32 attention kernels).  LayerNorm, adapter-sized
 *                           GEMMs and every row kernel stay exact fp32.  Host-side precision name: "fp16x3".
 *                           2 ("fp16x3f"): the forward pass as in 1 -- logits, token-keep decisions, losses and saved activations
 *                           are those of value 1 bit for bit -- while every GRADIENT product takes the hi*hi term alone (the
 *                           frozen-weight dgrad GEMMs contract dY_hi * W_hi; the attention backward's dP / dQ / dK / dV likewise, its
 *                           score recomputation keeps three products): gradients within 7e-4 relative L2 of the fp32 oracle
 *                           (value 1: 1.5e-4; the bar of the exact mode's test: 2e-3) at 0.81x the step time of value 1.
 *                           3 ("fp16x3h"): the forward as in 1 bit for bit, with what the backward pass reads saved in the 16-bit
 *                           operand type next to / instead of the fp32 tensors (q / k / v as hi + lo planes from the QKV epilogue, the
 *                           attention output, u, gelu'(z), the MLP output and the adapter bottleneck from their producers; ReLU /
 *                           dropout / gate masks are the exact forward's), and the BACKWARD pass on the 16-bit mode's data flow and
 *                           kernels (gradient operands x 2^12): gradients within 1.4e-3 of the oracle over five seeds, 0.85x the
 *                           step time of value 2.
 *                           4 ("fp16f8"): as 3, every forward GEMM as hi * hi on the f16 matrix cores + the two correction products
 *                           hi * lo + lo * hi on the fp8 (e4m3) ones (v_mfma_scale_f32_16x16x128_f8f6f4, twice the f16 rate per k: a
 *                           2K- instead of a 3K-equivalent contraction; operand images [hi16 | e4m3 hi | e4m3 lo 2^12], weights with a
 *                           per-matrix power of two, descaled by the MFMA's E8M0 scale operand): per GEMM 2e-5 of max|C| (three-part
 *                           1-2e-6, plain half 4e-4), logits ~5e-5, gate logits ~5e-5 from the fp32 reference -- inside the 1e-3 logit
 *                           bar, but token-keep decisions within ~1e-5 of the threshold can differ.
 *                           5 ("fp16x3q"): as 3 with only the attention branch's GEMMs (qkv, proj) in the form of 4, the MLP's
 *                           three-part: gate logits ~1e-5 (0 differing decisions over five seeds at B=16 and at B=128).  A complete_model
 *                           (teacher) pass takes the form of 4 for its MLP as well and the hi * hi product alone in its attention
 *                           forward -- its gate output is discarded, so no token-keep decision depends on it, only its logits
 *                           (<= 1e-4 from the reference at B=16, 2.7e-4 from the fp32 mode at B=128, instead of 7e-6).
 *                           Values >= 1 allocate a second arena (the [hi | lo] weight images, split-operand scratch, the 16-bit
 *                           tensors of 3..5) the first time they are set; plain fp32 contexts do not carry it. */
#define DYT_OPT_F32_SPLIT16 8
/*   DYT_OPT_ATTN_V2             bit mask, default 3 (environment DYT_ATTN_V2 overrides the default).  16-bit modes: bit 0 = the round-5
 *                           attention FORWARD kernel (csrc/attention_v2.hip: K / V / Q as swizzled LDS images filled by LDS-DMA from a loader
 *                           wave, V^T fragments by ds_read_b64_tr_b16 from the row-major image, online softmax per 32-key tile), bit 1 = the
 *                           round-5 BACKWARD kernel (four DMA images, every transposed operand a transposed read of the same image, row
 *                           statistics by the loader wave, results stored as whole rows from inside the next tile loop).  0: the round 1-4
 *                           kernels of csrc/attention.hip (DYT_OPT_ATTN_BWD_FUSED then selects among those).  Same arithmetic contract;
 *                           different summation order, so results agree to rounding, not bit for bit.  PROCESS-wide. */
#define DYT_OPT_ATTN_V2 9
/*   DYT_OPT_GEMM_SPLITK         default 1 (environment DYT_SPLITK overrides the default).  16-bit kernels: the K = 3072 GEMMs of the cls-only
 *                           last block (DYT_OPT_CLS_TAIL; M = batch rows: fc2 forward with the adapter pair, fc1 dgrad) run split over
 *                           256-wide k slices on ~300 workgroups + one reduce launch that applies the epilogue, instead of 6 tiles of
 *                           128x128 (csrc/gemm_skinny.h).  Slices are summed in a fixed order: deterministic; agrees with the tile kernels to
 *                           fp32 rounding of the k sums, not bit for bit.  PROCESS-wide. */
#define DYT_OPT_GEMM_SPLITK 10
/*   DYT_OPT_LEARNABLE_SCALE     default 0.  1: tuning_config.ffn_adapter_scalar == "learnable_scalar" (models/dynamic_adapter.py:101-102, 138: the
 *                           adapter output is multiplied by a trainable nn.Parameter(torch.ones(1)) per block instead of a constant).  The
 *                           parameter is DYT_P_AD_SCALE of the flat buffer (one word per block, behind the gate bias; set it like any
 *                           trainable tensor), dyt_config.adapter_scale is ignored.  The per-step adapter copies carry the scale
 *                           (W' = s W_up, b' = s b_up), so every kernel runs as with scale 1; the backward leaves dL/dW', dL/db' in a
 *                           scratch buffer and one small kernel applies the chain rule: dW_up = s dW', db_up = s db',
 *                           ds = <dW', W_up> + <db', b_up>.  The scale words belong to the caller's flat buffer and are whatever it holds:
 *                           write the reference's initial value 1.0 (models/dynamic_adapter.py:102) before the first pass -- a zeroed
 *                           buffer means s = 0, i.e. adapters off and no up-projection gradient.  Must be set before the first forward
 *                           pass of the context; changing it afterwards returns DYT_ERR_STATE. */
#define DYT_OPT_LEARNABLE_SCALE 11
int dyt_ctx_set_option(dyt_ctx* ctx, int option, int value);
/* the process-wide options (DYT_OPT_ATTN_BWD_FUSED, DYT_OPT_ATTN_V2, DYT_OPT_GEMM_SPLITK) without a context: unit entries such as dyt_attention() see them too */
int dyt_set_global_option(int option, int value);

/* Stochastic depth -- timm's DropPath as the reference's blocks use it (models/vision_transformer_IN21K.py:121,131 construct it with
 * dpr[l] = linspace(0, drop_path_rate, depth)[l], :285; :148 x + drop_path1(attn(norm1 x)), :159 drop_path2(mlp(norm2 x));
 * main_image.py:118,213 --drop_path).  In every TRAINING forward pass (DYT_F_TRAINING) block l > 0 multiplies the attention branch and
 * the MLP branch of image b by two independent factors: 1 / keep_l with probability keep_l = 1 - rate * l / (depth - 1), else 0
 * (scale_by_keep); evaluation passes and rate 0 (the default, what every reference script runs) do nothing.  The factors ride in the
 * residual epilogues of the proj and fc2 GEMMs; the backward pass scales the two branch gradients by the same factors.  Each pass
 * (student, complete_model) draws its own, from the call's Philox seed (sub-stream 0x10000 + 2 l + branch of the slot).
 * With a rate > 0 the 16-bit modes run the adapter's up-projection as its own launch again (DYT_OPT_FC2_CAT has no scaled form). */
int dyt_set_drop_path(dyt_ctx* ctx, float rate);
/* Tests / reproducibility across frameworks: the factors of the next training passes of `slot` are read from `scales` (device,
 * [2][depth][batch of the call]: [0] attention branch, [1] MLP branch; row 0 = block 0 is ignored, it is never dropped) instead of being
 * drawn; the pointer is kept, not copied; null restores the library's own draws. */
int dyt_set_drop_path_scales(dyt_ctx* ctx, int slot, const float* scales);

/* Copy one FROZEN parameter (fp32, reference state_dict layout) into the context; the library
 * keeps its own copies in the layouts/dtypes its kernels want (incl. transposes for dgrad).
 * Replaces load_state_dict(...) for the frozen keys -- main_image.py:245. */
int dyt_set_frozen(dyt_ctx* ctx, int param, int layer, const float* src, void* stream);

/* The 74 trainable tensors live in ONE flat fp32 buffer owned by the caller (so one AdamW
 * launch and one all-reduce cover them).  Offsets in elements. -- main_image.py:250-256,285 */
int dyt_trainable_numel(const dyt_ctx* ctx, int64_t* numel);
int dyt_trainable_offset(const dyt_ctx* ctx, int param, int layer, int64_t* offset, int64_t* numel);

/* VisionTransformer.forward(x, complete_model) -- models/vision_transformer_IN21K.py:343-385,
 * Block.forward :144-165, TokenSelect.forward / _gumbel_sigmoid dynamic_adapter.py:25-77,
 * Adapter.forward :120-140.
 *   images        [B,3,224,224] fp32
 *   trainable     flat trainable buffer (dyt_trainable_offset layout)
 *   g1, g2        [depth,B,196] fp32 Gumbel draws to inject (parity tests), or NULL: logistic noise
 *                 from the on-device Philox stream (seed, slot) -- same distribution as g1-g2
 *   keep_mask     [depth,B*197,r] uint8 adapter-dropout keep mask to inject, or NULL: Philox
 *   logits        [B,C] out
 *   token_select  [B,depth,196] out fp32 {0,1} (may be NULL)
 *   token_logits  [B,depth,196] out fp32       (may be NULL)
 */
int dyt_forward(dyt_ctx* ctx, int slot, const float* images, int batch, int flags,
                const float* trainable, const float* g1, const float* g2,
                const uint8_t* keep_mask, uint64_t seed,
                float* logits, float* token_select, float* token_logits, void* stream);

/* Backward of one saved pass: what loss.backward() does through the module
 * (engine_finetune.py:74-76 via misc.py:258-259).  Gradients of the trainable tensors are
 * ACCUMULATED into grad_flat (same layout as `trainable`).
 *   dlogits        [B,C]
 *   dtoken_select  [B,depth,196] or NULL  (gradient w.r.t. the straight-through mask)
 *   dtok_uniform   device float[3] or NULL: {uniform, extra for dropped, extra for kept} gradient per
 *                  mask element, as produced by dyt_loss (used when dtoken_select is NULL)
 *   dtoken_logits  [B,depth,196] or NULL
 */
int dyt_backward(dyt_ctx* ctx, int slot, const float* dlogits, const float* dtoken_select,
                 const float* dtok_uniform, const float* dtoken_logits, float* grad_flat, void* stream);

/* Loss of the fine-tune step and its gradient w.r.t. both logits -- engine_finetune.py:52-63 and
 * AdaLoss.forward models/losses.py:48-84:
 *   loss = CE(s,y) + ratio*((mean(mask)-target)^2 + w_min*sum(clamp(t_min-mask,0))) + CE(t,y)
 *          + KL(log_softmax s || log_softmax t.detach(), batchmean)
 * The mask statistics are taken from the token counts saved by the student pass in `slot_student`.
 *   out_losses  device float[8]: loss, base_loss, token_loss(scaled), teacher_loss, distillation_loss,
 *               mean keep ratio, kept tokens, 0
 *   dtok        device float[3] (see dyt_backward)
 */
int dyt_loss(dyt_ctx* ctx, int slot_student, const float* logits_s, const float* logits_t,
             const int64_t* targets, int batch, float token_target_ratio, float token_loss_ratio,
             float token_minimal, float token_minimal_weight,
             float* dlogits_s, float* dlogits_t, float* out_losses, float* dtok, void* stream);

/* Mixup / soft labels (reference engine_finetune.py:44-45,60-62: `samples, targets = mixup_fn(samples, targets)` hands class-probability
 * targets [rows, num_classes] to nn.CrossEntropyLoss for the student and the teacher pass alike): the loss evaluations that follow -- dyt_loss,
 * dyt_step_fwd_bwd -- read `targets` (device fp32; the pointer is kept, not copied; rows = the logits' rows) instead of their integer labels,
 * which are then ignored: CE = -sum_c t_c log p_c, d logits = p sum_c t_c - t.  NULL restores the integer labels. */
int dyt_set_soft_targets(dyt_ctx* ctx, const float* targets, int rows);

/* torch.optim.AdamW over the flat trainable buffer -- main_image.py:285; the gradient is first
 * multiplied by grad_scale (1/world_size after a SUM all-reduce).  `step` is 1-based. */
int dyt_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
              int step, float lr, float beta1, float beta2, float eps, float weight_decay,
              float grad_scale, void* stream);

/* The same update guarded like the reference's NativeScaler / GradScaler.step (misc.py:256-272; the reference trains under fp16 autocast):
 * when grad holds an inf or NaN -- a 16-bit operand overflowed somewhere in the step -- parameters and moments are left untouched and
 * the skip is counted, without a host round trip.  state = device int32[4], zero-initialised and owned by the caller's optimizer:
 * {updates applied, updates skipped, non-finite flag of this call, reserved}; the bias corrections use state[0] + 1 as the step. */
int dyt_adamw_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, int32_t* state,
                      float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);

/* One whole fine-tune step body (engine_finetune.py:47-79 without the host syncs): student +
 * teacher forward, loss, one backward over both passes into grad_flat (zeroed first).  The caller
 * all-reduces grad_flat (DDP, main_image.py:280-282) and then calls dyt_adamw. */
int dyt_step_fwd_bwd(dyt_ctx* ctx, const float* images, const int64_t* targets, int batch,
                     int flags, const float* trainable,
                     const float* g1, const float* g2, const uint8_t* keep_mask, uint64_t seed,
                     float token_target_ratio, float token_loss_ratio,
                     float token_minimal, float token_minimal_weight,
                     float* grad_flat, float* out_losses, float* logits_s, float* logits_t,
                     float* token_select, void* stream);

/* Set the device-side seed word used by DYT_F_DEVICE_SEED passes (one tiny launch on `stream`). */
int dyt_seed(dyt_ctx* ctx, uint64_t seed, void* stream);

/* Chunked all-reduce (what DDP's bucketed hooks do inside loss.backward(), misc.py:258-259 / main_image.py:280-282):
 * the backward pass runs block 11 -> 0, so part 0 of the flat gradient = the head and blocks >= depth/2 (a contiguous
 * tail of the buffer) is final while the frozen-backbone backward of the lower blocks is still running.
 *   dyt_grad_part(ctx, part, &offset, &numel)   part 0: early (upper) range, part 1: the rest
 *   dyt_stream_wait_grads(ctx, 0, comm_stream)  makes comm_stream wait (device-side) until part 0 of the last
 *                                               dyt_step_fwd_bwd is complete, so its all-reduce overlaps the rest of
 *                                               the backward; the step's own stream owns the whole buffer once
 *                                               dyt_step_fwd_bwd's work is done, as before. */
int dyt_grad_part(const dyt_ctx* ctx, int part, int64_t* offset, int64_t* numel);
int dyt_stream_wait_grads(dyt_ctx* ctx, int part, void* stream);

/* The gradient all-reduce itself, on RCCL (replaces DistributedDataParallel's bucket all-reduce, main_image.py:280-282; the
 * collective call sites of SURVEY.md section 2.3): SUM of grad_flat over the ranks of `rccl_comm` (an ncclComm_t the binder created
 * with ncclCommInitRank; one process per GPU).  With a `comm_stream` different from `stream`, part 0 is reduced on comm_stream as soon
 * as the last dyt_step_fwd_bwd has it final (overlapping the backward of the lower blocks), part 1 on `stream`, and `stream` owns the
 * whole buffer afterwards; comm_stream = NULL: one all-reduce on `stream`.  The 1/world factor is the caller's (dyt_adamw's
 * grad_scale).  RCCL is bound at load time from the host process (no link-time dependency): DYT_ERR_STATE when it is absent. */
int dyt_allreduce_grads(dyt_ctx* ctx, void* rccl_comm, float* grad_flat, void* comm_stream, void* stream);

/* torch.nn.utils.clip_grad_norm_ over the flat gradient (engine_finetune.py:74 -> misc.py:262-266, --clip_grad):
 * norm_out[0] (device, may be NULL) = || pre_scale * grad ||_2 ; grad *= min(1, max_norm / (norm + 1e-6)).
 * pre_scale is the factor AdamW will apply (1/world after a SUM all-reduce, 1/accum_iter). */
int dyt_clip_grad_norm(dyt_ctx* ctx, float* grad, int64_t numel, float max_norm, float pre_scale, float* norm_out, void* stream);

/* Test hook: the token dispatcher's index arrays of a pass still held in `slot` (compacted student pass), block `layer`:
 *   row_src[K] compact row -> token row (ascending = nonzero() of model_speed_test.py:300), dst_of[B*197] token row ->
 *   compact row or -1, counts[B] kept tokens per image incl. cls, total[1] = K.  Device-to-device copies on `stream`;
 *   any pointer may be NULL. */
int dyt_debug_dispatch(dyt_ctx* ctx, int slot, int layer, int32_t* row_src, int32_t* dst_of, int32_t* counts, int32_t* total,
                       void* stream);
/* Test accessor: the adapter bottleneck relu(down(u)) (x dropout scale) a saved pass holds for `layer`, as fp32 [rows, 64] (64 = the
 * padded rank); *rows_out = B*197, or B when the last block ran in the cls-only tail form.  out: device memory for B*197*64 floats. */
int dyt_debug_dact(dyt_ctx* ctx, int slot, int layer, float* out, int* rows_out, void* stream);
/* tests: the stochastic-depth factors [2][depth][batch] the last pass of `slot` ran with (drawn or injected); DYT_ERR_STATE when it ran without */
int dyt_debug_drop_path(dyt_ctx* ctx, int slot, float* out, void* stream);

/* ---- sub-module entry points: they allocate scratch and synchronise; not for the hot loop ---- */
/* Adapter.forward (models/dynamic_adapter.py:120-140, layernorm option "none"): out[M,768] = [residual +] scale *
 * (dropout_p(relu(x down_w^T + down_b)) up_w^T + up_b).  down_w [r,768], up_w [768,r] (nn.Linear layouts), r <= 64; residual may be
 * NULL (add_residual=False); drop_p > 0: keep_mask [M,r] (uint8) if given, else Philox(seed). */
int dyt_adapter_fwd(const float* x, const float* down_w, const float* down_b, const float* up_w, const float* up_b,
                    const float* residual, float* out, int M, int r, float scale, float drop_p, const uint8_t* keep_mask,
                    uint64_t seed, int precision, void* stream);
/* its backward for dout [M,768] with the same draws: dx [M,768] (may be NULL); the parameter gradients are ACCUMULATED */
int dyt_adapter_bwd(const float* x, const float* down_w, const float* down_b, const float* up_w, const float* dout, float* dx,
                    float* d_down_w, float* d_down_b, float* d_up_w, float* d_up_b, int M, int r, float scale, float drop_p,
                    const uint8_t* keep_mask, uint64_t seed, int precision, void* stream);
/* the token-gathered MLP of block `layer` (frozen weights of the context; models/model_speed_test.py:297-305): for the tokens with a
 * non-zero mask, x[t] += fc2(gelu(fc1(LN2(u[t])))); u, x fp32 [batch*197,768] (x in place), mask [batch*197]; total_out[1] (device,
 * may be NULL) = number of gathered tokens */
int dyt_mlp_gathered_fwd(dyt_ctx* ctx, int layer, const float* u, const float* mask, float* x, int batch, int32_t* total_out,
                         void* stream);
/* Its backward (dyt_mlp_gathered_bwd): for dy = the gradient w.r.t. x above, du [B*197,768] += the gradient that reaches u
 * through the gathered MLP of block `layer` -- LN2 backward of fc1^T (gelu'(z) * (fc2^T dy)) scattered back to the kept tokens, nothing
 * for the dropped ones (the residual path du += dy is the caller's; frozen weights: no weight gradients).  Autograd of
 * models/model_speed_test.py:297-305.  Unit-test entry: recomputes the forward, allocates scratch and synchronises. */
int dyt_mlp_gathered_bwd(dyt_ctx* ctx, int layer, const float* u, const float* mask, const float* dy, float* du, int batch, void* stream);

/* One nn.Linear (c = a w^T + bias; a [M,K], w [N,K], bias [N] or NULL, c [M,N], fp32 device pointers) through the split forms of the
 * fp32 mode: form 3 = every product as three IEEE-half products (hi*hi + hi*lo + lo*hi), form 8 = hi*hi on the f16 matrix cores plus
 * the two correction products on the fp8 (e4m3) matrix cores ("fp16f8").  K % 128 == 0, N % 128 == 0.  Unit-test entry: allocates
 * scratch and synchronises.  Reference op: nn.Linear of models/vision_transformer_IN21K.py:56,73 and timm Mlp (:124-129). */
int dyt_linear_split(const float* a, const float* w, const float* bias, float* c, int M, int N, int K, int form, void* stream);

/* ---- single-kernel entry points (unit tests; also the building blocks of the sub-module API) ---- */
/* nn.LayerNorm(768, eps=1e-6) forward; out fp32 */
int dyt_layernorm(const float* x, const float* w, const float* b, float* out, int rows, void* stream);
/* C[M,N] = A[M,K] @ W[N,K]^T + bias  (nn.Linear); precision selects the kernel family */
int dyt_linear(const float* a, const float* w, const float* bias, float* c, int M, int N, int K,
               int precision, void* stream);
/* scaled_dot_product_attention over [B,12,197,64] from a fused qkv [B*197,2304] (Attention.forward
 * vision_transformer_IN21K.py:54-72): out [B*197,768]; if dout != NULL also returns dqkv */
int dyt_attention(const float* qkv, float* out, const float* dout, float* dqkv, int batch,
                  int precision, void* stream);
/* TokenSelect.forward + index compaction for one block: u [B*197,768], w[768], b[1];
 * outputs mask [B,196], logits [B,196], keep_idx int32 [B*197] (flat kept rows, ascending),
 * counts int32 [B] (kept per image incl. cls), total int32[1] */
int dyt_gate_compact(const float* u, const float* w, const float* b, const float* g1, const float* g2,
                     int batch, int training, float tau, float threshold,
                     float* mask, float* logits, int32_t* keep_idx, int32_t* counts, int32_t* total,
                     void* stream);

/* ---- measurement hooks (bench.py, tools/gemm_bench.py) ---- */
/* C[M,N] (bf16) = A[M,K] (bf16) @ W[N,K]^T (bf16), fp32 accumulate; `variant` selects a kernel build
 * (0 / 10 / 70: the 128x128, 256x256 and pre-shuffled-weight kernels, 30: the product dispatch, 9 / 19 / 79: with phase timers) */
int dyt_gemm_bf16_raw(const void* a, const void* w, void* c, int M, int N, int K, int variant, void* stream);
/* C[M,N] (fp32) = A[M,K] (fp32) @ W[N,K]^T (fp32) on the exact-fp32 MFMA kernel of the parity mode (csrc/gemm_f32_mfma.h);
 * N % 64 == 0, K % 64 == 0; no synchronisation.  variant 0: plain store; 1: the adapter up-projection epilogue (residual read
 * from c + M*N, c = resid + 0.1 * acc); 2: accumulate (c += acc) */
int dyt_gemm_f32_raw(const float* a, const float* w, float* c, int M, int N, int K, int variant, void* stream);
/* the adapter weight-gradient kernel alone: out_w[c*r + j] += sum_m X[m][c] Y[m][j] (X [M,768], Y [M,64], fp32 or bf16 by
 * `precision`), out_xsum[c] += sum_m X[m][c], out_ysum[j] += sum_m Y[m][j]; `partial` = dyt_wgrad_scratch_floats(M) floats */
int64_t dyt_wgrad_scratch_floats(int M);
int dyt_wgrad_raw(const void* X, const void* Y, int M, int r, int precision, float* partial, float* out_w, float* out_xsum,
                  float* out_ysum, void* stream);
/* measurement hooks of tools/probes/determinism_*.py: with DYT_DBG_CKSUM=1 in the environment every backward
 * launch is followed by an integer checksum of its output; read back per pass (slot 0 student, 1 teacher) with its label; one
 * launch's output can be kept whole (DYT_DBG_DUMP="<slot>:<label>") and copied out.  No effect unless the variables are set. */
int dyt_debug_checksums(int slot, uint64_t* out, int max_n, int* n_out);
const char* dyt_debug_checksum_label(int slot, int i);
int64_t dyt_debug_dump_read(void* dst_device, int64_t max_bytes);
/* phase timers of the instrumented GEMM variants: cycles {prologue, main loop, epilogue} summed over
 * workgroups and the workgroup count; synchronises; optionally resets */
int dyt_debug_counters(uint64_t* out4, int reset);
/* bracket every kernel launch with hipEvents on `stream` and accumulate per category */
int dyt_profile_enable(dyt_ctx* ctx, int on);
/* categories: 0 gemm (big MFMA GEMMs), 1 attention, 2 everything else.  Returns accumulated
 * milliseconds, launches and algorithmic FLOPs since the last call, then resets.  Synchronises.
 * category 3: `launches` = bf16 GEMM KERNEL launches since dyt_profile_enable(ctx, 1) (one GEMM of category 0 may be
 * two kernel launches with different tiles); ms / flops are 0; no reset, no synchronisation. */
int dyt_profile_read(dyt_ctx* ctx, int category, double* ms, int64_t* launches, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* DYT_HIP_H */
