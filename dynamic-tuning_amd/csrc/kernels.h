// Host-callable kernel launchers of libdyt_hip.so (internal API between translation units).
#pragma once
#include <stdlib.h>
#include "dyt_common.h"

namespace dyt {

// ------------------------------------------------------------------------------------------
// GEMM family: C[M,N] = A[M,K] @ W[N,K]^T with a fused epilogue.  A and W are K-contiguous
// (nn.Linear layout); dgrad uses a pre-transposed copy of the frozen weight, so every dense
// contraction on the path is this one "NT" form.
// ------------------------------------------------------------------------------------------
enum EpiKind {
    EPI_BIAS_F32 = 0,   // out_f32 = acc + bias                                   (generic nn.Linear)
    EPI_QKV,            // +bias, q*0.125, scatter to q/k/v [B,12,197,64]          (Attention.qkv)
    EPI_BIAS_RESID,     // out_f32 = acc + bias + resid ; optional AT copy          (Attention.proj + residual)
    EPI_FC1,            // z = acc + bias ; gelu(z) -> out_at ; gelu'(z) -> out_at2 (optional, for backward) (Mlp.fc1 + GELU)
    EPI_FC2,            // h = acc + bias ; x[row_map[r]] += mask*h ; optional h save (Mlp.fc2 + scatter + residual)
    EPI_GELU_BWD,       // out_at = acc * aux (aux = gelu'(z) saved by EPI_FC1)        (dgrad through fc2, GELU)
    EPI_STORE_F32,      // out_f32 (+)= acc
    EPI_STORE_AT,       // out_at = acc
    EPI_AD_DOWN,        // out_at = dropout(relu(acc + bias))                       (Adapter.down_proj)
    EPI_AD_UP,          // out_f32 = resid + scale*(acc + bias)                     (Adapter.up_proj * scale + residual)
    EPI_AD_DGRAD_UP,    // out_at = acc * scale * relu'/dropout mask                (dgrad through up_proj)
    EPI_EMBED,          // x0[b*197+1+p] = acc + bias + pos[1+p]                    (PatchEmbed + pos_embed)
    EPI_BIAS_AT,        // out_at = acc + bias (bias may be null)                    (video pooling head k / v projections)
};

struct GemmArgs {
    const void* A = nullptr;      // [M,K]  AT
    const void* W = nullptr;      // [N,K]  AT
    const void* Wp = nullptr;     // optional: the same (frozen) weight pre-shuffled into MFMA fragment order (gemm_bpre.h)
    int M = 0, N = 0, K = 0;
    const int* m_dev = nullptr;   // device-side count of valid rows (<= M), or null
    const int* a_map = nullptr;   // gather: logical A row r is read from A[a_map[r]] (null = identity)
    // K-concatenated form (16-bit kernels, EPI_FC2): C = A2 W2^T + A W^T, A2 [rows,64] (row r read from A2[a2_map[r]]), W2 [N,64];
    // bias2 / scale: the second pair's bias enters as scale * bias2 (A2 is expected to carry `scale` already)
    const void* A2 = nullptr; const void* W2 = nullptr; const int* a2_map = nullptr; const float* bias2 = nullptr;
    // fp32 mode, "split" form: W3 = the fp32 weight as a [N, 2K] 16-bit image [hi | lo] (launch_split3_w); the fp32 A likewise as a3
    // [M, 2K] = [hi | lo] (written by its producer or split on the fly) and the GEMM runs as ONE 16-bit MFMA contraction over 3K k-tiles
    // (hi*hi + hi*lo + lo*hi: the loaders fold the tile index onto the two stored parts)
    // with the fp32 epilogue: the accuracy of the exact-fp32 MFMA kernel at ~2.7x its speed (tools/probes/split_precision_probe.py)
    const void* W3 = nullptr; void* a3 = nullptr;
    float a3_scale = 1.f;   // power of two the fp32 A is multiplied by before it is split (gradients: keeps the lo part out of the fp16
                            // subnormals); the accumulators are multiplied by out_scale = 1 / a3_scale before the epilogue functor
    float out_scale = 1.f;
    int a3_parts = 3;        // products of the split contraction (3: full; 2 / 1: gradient GEMMs of the forward-parity variant)
    int a_ld = 0;            // (internal) row stride of the split operands
    int a_fold = 0;          // (internal) k-tiles per part of a split A operand stored [hi | lo]: set by launch_gemm
    int f8_begin = 0;        // (internal) F8 kernels: index of the first fp8 k-tile
    bool a3_ready = false;   // a3 already holds the split A (written by the producing kernel): no pre-pass
    bool a3_mapped = false;  // ... one split row per SOURCE row: the GEMM gathers through a_map (the pre-pass compacts instead)
    // FC1 / GELU_BWD in the split form: the result also (instead of out_at) goes out as the split A operand of the NEXT GEMM
    void* out3 = nullptr; float out3_scale = 1.f; bool out3_hi_only = false;   // hi_only: the next GEMM contracts the hi part alone
    // fp32 mode whose backward runs on 16-bit operands ("fp16x3h"): the outputs that exist only for the backward pass -- BIAS_RESID's
    // out_at (copy of u), FC1's out_at2 (gelu'), FC2's h_out, AD_DOWN's out_at2 (copy of d_act) -- are of the 16-bit operand type
    bool save16 = false;
    // "fp16f8" form of the split contraction: operands in the hi16 / fp8 images (dyt_common.h: store4_split_f8), the correction
    // products on the fp8 matrix cores; w_exp = the weight image's device-side exponent word
    bool f8 = false; const int* w_exp = nullptr;
    void* qkv_lo[3] = {nullptr, nullptr, nullptr};   // QKV in the split forms: q / k / v as 16-bit hi planes (out_at, out_at2, out_at3) + these lo planes
    bool out3_f8 = false;   // FC1: out3 (the fc2 GEMM's operand) in that form
    // LayerNorm folded into the GEMM behind it (16-bit modes): BIAS_RESID (the producer of the row u) also writes per-row
    // partial statistics -- (sum, centred sum of squares) of each 64-column group, [M][LN_PARTS] -- and FC1 (the consumer) contracts the
    // 16-bit copy of the UN-normalised row with W' = AT(gamma W) and applies z = rstd (acc - mean colsum(W')) + (W beta + b) in its
    // epilogue.  FC1 takes the partials (ln_part, indexed by source row: a_map), merges them itself -- in the kernel's prologue, or by a
    // small pre-pass into ln_scratch [M] for the tile shapes without that prologue -- and leaves (mean, rstd) per source row in ln_st_out
    float2* ln_part = nullptr; float2* ln_st_out = nullptr; float2* ln_scratch = nullptr; const float* ln_cs = nullptr;
    const float* bias = nullptr;
    float* out_f32 = nullptr;
    void* out_at = nullptr;
    void* out_at2 = nullptr;
    void* out_at3 = nullptr;
    const float* resid = nullptr;     // BIAS_RESID / AD_UP residual ; FC2: residual source when not in place (null = out_f32)
    const void* aux_at = nullptr;     // z (GELU_BWD) / d_act (AD_DGRAD_UP)
    const int* row_map = nullptr;     // FC2 / AD_UP scatter, AD_DOWN mask index, GELU_BWD aux index: compact row -> token row
    const float* row_mask = nullptr;  // FC2 masked-dense: per-token mask ; AD_UP: rows with mask != 0 are skipped
    void* h_out = nullptr;            // FC2: save h (AT)
    const uint8_t* keep = nullptr;    // AD_DOWN injected keep mask [M, r]
    const float* pos = nullptr;       // EMBED
    int r = 0;                        // adapter rank (columns >= r of the padded bottleneck are dead)
    float scale = 1.f;
    float bias_scale = -1.f;          // AD_UP: the bias enters as bias_scale * bias instead of scale * bias (>= 0: set; the operand then carries the adapter scale itself)
    float inv_keep = 1.f;             // 1/(1-p) when training, else 1
    float drop_p = 0.f;               // >0 : apply dropout (Philox when keep == null)
    uint64_t seed = 0, subseq = 0;
    const uint64_t* seed_dev = nullptr;   // when set, the Philox seed is read from this device word (graph replay)
    int accumulate = 0;
    // stochastic depth (timm DropPath, reference models/vision_transformer_IN21K.py:121,131,148,159): per-IMAGE factors (0 or 1 / keep) on the
    // branch a residual epilogue adds -- BIAS_RESID / mapped AD_UP: x + rs[b] (acc + bias); FC2: u + rs[b] mask (acc + bias) -- b = token row / 197
    const float* row_scale = nullptr;
    // skinny K >= 1024 GEMMs (the cls-only last block): workspace for the split-K form (gemm_skinny.h), [slices][M][N] fp32; null = tile kernels
    float* splitk_ws = nullptr; size_t splitk_ws_bytes = 0;
    double flops() const { return 2.0 * M * (double)N * K; }
};

int launch_gemm(int precision, EpiKind kind, const GemmArgs& a, hipStream_t s);
// fc1 of the folded form, from the fp32 weight W [N,768]: Wf = AT(gamma W) (+ its fragment-order twin when Wfp), cs[n] = sum_k Wf[n,k],
// bf[n] = bias[n] + sum_k W[n,k] beta[k]
int launch_ln_fold_w(int precision, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, void* Wfp, float* cs,
                     float* bf, int N, hipStream_t s);
// fp32 W [N,K] -> [N, 2K] 16-bit image [hi | lo] with hi = rn16(w), lo = rn16(w - hi)
int launch_split3_w(const float* W, void* W3, int N, int K, hipStream_t s);
// ... in the hi16 / fp8 form [N, hi16 | e4m3(lo 2^(ew+11)) | e4m3(hi 2^ew)], ew = 7 - ceil(log2 max|w|) written to ew_dev[0]; scratch: one device word
int launch_split_w_f8(const float* W, void* W3, int N, int K, int* ew_dev, unsigned* scratch, hipStream_t s);
// bf16 W [N,K] row-major -> fragment order for the pre-shuffled-weight kernel (N % 16 == 0, K % 32 == 0)
int launch_preshuffle_w(const void* W, void* Wp, int N, int K, hipStream_t s);
int gemm_debug_counters(unsigned long long* out4, int reset);
long long gemm_kernel_launch_count(int reset);
int launch_gemm_raw(const void* A, const void* W, void* C, int M, int N, int K, int variant, hipStream_t s);
int launch_gemm_f32_raw(const void* A, const void* W, void* C, int M, int N, int K, int variant, hipStream_t s);

// ------------------------------------------------------------------------------------------
// attention (N = 197, d = 64, 12 heads); q pre-scaled by 1/8 in the QKV epilogue
// ------------------------------------------------------------------------------------------
// q,k,v: [B*12][197][64] AT ; out: [B*197][768] AT ; lse: [B*12][197] f32
// 16-bit copies of q / k / v (same layout) and of the output [B*197][768] written by the split forward kernel for a 16-bit backward
struct AttnSave16 { void* q = nullptr; void* k = nullptr; void* v = nullptr; void* o = nullptr;
                    // planes: q / k / v above are INPUT hi planes (written by the QKV epilogue) and these the lo planes; the fp32 q / k / v arguments are unused
                    const void* q_lo = nullptr; const void* k_lo = nullptr; const void* v_lo = nullptr; };
int launch_attn_fwd(int precision, const void* q, const void* k, const void* v, void* out, float* lse,
                    int batch, hipStream_t s, int split16 = 0, void* out3 = nullptr, const AttnSave16* save16 = nullptr, int out3_f8 = 0,
                    int parts = 3);   // parts = 1 (split kernel on planar q / k / v): S and P V as the hi * hi product alone   // out may be null when out3 is given   // out3: + the output as a [M][3*768] split operand   // split16 (fp32 mode): products as three 16-bit MFMA products
void set_attn_f32_split(int on);   // process-wide version of split16 (unit entries)
// dqkv: [B*197][2304] AT (dq already multiplied by 1/8) ; delta: scratch [B*12][197] f32
int launch_attn_bwd(int precision, const void* q, const void* k, const void* v, const void* out,
                    const void* dout, const float* lse, float* delta, void* dqkv, int batch, hipStream_t s, int q_tiles = 7,
                    void* dqkv3 = nullptr, float s3 = 1.f, int split16 = 0, int grad_parts = 3, int out_hi_only = 0, int out_ld = 0);   // out_ld: row stride of `out` (0 = 768; 16-bit kernels: the hi plane of a split operand image)   // out_hi_only: dqkv3 without its lo half (the qkv dgrad contracts one part)
      // split16 (fp32 mode): the split backward kernels; grad_parts = 1: their dP / dQ / dK / dV products as hi * hi alone   // dqkv3 (fp32 mode): dqkv * s3 as split 16-bit operand [M, 3 * 2304] instead of fp32
// q_tiles: 32-row query tiles that can carry a non-zero dout (1: only the cls rows do); honoured by the fused 16-bit kernel, exact
// 16-bit modes: 1 (default) = dQ and dK/dV of a head in one persistent kernel, 0 = the two separate kernels (process-wide)
void set_attn_bwd_fused(int on);
// round-5 kernels (attention_v2.hip; 16-bit modes): bit 0 = forward (online softmax, LDS-DMA images, two workgroups per CU), bit 1 = backward
void set_gemm_splitk(int on);   // DYT_OPT_GEMM_SPLITK
int get_gemm_splitk();
void set_attn_v2(int mask);
int get_attn_v2();
int launch_attn_fwd_v2(const void* q, const void* k, const void* v, void* out, float* lse, int batch, hipStream_t s, int out3_f8 = 0);   // out3_f8: `out` is the proj GEMM's operand image in the hi16 / fp8 form (rows of SPLIT_A * 768)
int launch_attn_bwd_v2(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, void* dqkv,
                       int batch, hipStream_t s, int q_tiles, int out_ld);

// ------------------------------------------------------------------------------------------
// row-wise / small kernels
// ------------------------------------------------------------------------------------------
// LayerNorm over 768 channels, one wave per row; stats[row] = {mean, rstd}
int launch_ln_fwd(int precision, const float* x, const float* w, const float* b, void* out, float2* stats,
                  int rows, hipStream_t s, void* out3 = nullptr, int out3_f8 = 0);   // out3 (fp32 mode): the rows as split 16-bit operand [rows, SPLIT_A*768] instead of out; out3_f8: in the hi16 / fp8 form (store4_split_f8)
int launch_ln_fwd_f32out(const float* x, const float* w, const float* b, float* out, int rows, hipStream_t s);
// dx_out[row] = base[row] + LNbwd(dy[row]; x[row], stats[row], w)
int launch_ln_bwd(int precision, const void* dy, const float* x, const float2* stats, const float* w, const float* base,
                  float* dx_out, int rows, void* g_at, const void* h_next, const int* dst_of_next, float* dmask_next,
                  float gs, hipStream_t s, void* out3 = nullptr, float s3 = 1.0f, int out3_hi_only = 0, const void* base_at = nullptr);   // base_at: the stream as a 16-bit gs-scaled operand copy instead of `base` (dx_out may then be null)   // out3: + dx * s3 as a [rows][3*768] split operand (fp32 mode)   // gs: factor carried by the 16-bit gradient operands dy (in) and g_at (out)

// the adapter's own LayerNorm (dyt_config::adapter_ln; reference models/dynamic_adapter.py:95-98,121-122,132-133; eps 1e-5): out = LN(x) w + b (+ resid),
// out of the 16-bit operand type (out_at_precision != 0) or fp32; stats[row] = (mean, rstd)
int launch_adapter_ln_fwd(int out_at_precision, const float* x, const float* w, const float* b, void* out, float2* stats, const float* resid,
                          int rows, hipStream_t s);
// out[0:768] += sum_t dy[t] xhat[t] (dgamma), out[768:1536] += sum_t dy[t] (dbeta); dy of the operand type (x gs) when precision != 0; fixed order
int launch_ln_param_grad(int precision, const void* dy, const float* x, const float2* stats, float* partial, float* out, int rows, float gs, hipStream_t s);
int64_t ln_param_grad_scratch_floats(int rows);

struct GateArgs {
    const float* u;          // [B*197,768] residual stream after attention
    const float* w;          // [768]
    const float* b;          // [1]
    const float* g1;         // [B,196] or null
    const float* g2;         // [B,196] or null
    int batch;
    int training;
    float tau, threshold;
    uint64_t seed, subseq;   // Philox stream when training && !g1
    const uint64_t* seed_dev = nullptr;   // when set, overrides `seed` with the device-side word (graph replay)
    float* soft;             // [B*197] y_soft per token (cls slot unused) -- saved for backward
    float* maskf;            // [B*197] hard mask per token as float (cls = 1)
    float* out_select;       // user tensor [B,depth,196] (pointer already offset to this layer) or null
    float* out_logits;       // idem
    int out_stride;          // depth*196
    int force_first = 0;     // > 0: keep exactly the tokens n < force_first (Block.forward_count_flops), ignore the gate
    int* keep_local;         // [B,197] kept token ids per image, ascending
    int* counts;             // [B]
};
int launch_gate(const GateArgs& a, hipStream_t s);
// LN2 of the kept rows into the compact A operand; row_src[dst]=src token row, dst_of[src]=dst or -1;
// also computes the row offsets (prefix of counts) itself and publishes total[0] = sum(counts)
int launch_ln_gather(int precision, const float* u, const float* w, const float* b, const int* keep_local,
                     const int* counts, int* total, const float* maskf, void* out, float2* stats,
                     int* row_src, int* dst_of, int batch, hipStream_t s, void* out3 = nullptr, int out3_f8 = 0,
                     int* drop_src = nullptr);   // drop_src: + the dropped tokens' rows, ascending, and total[1] = their number (as launch_gather_index)

// the index half of launch_ln_gather alone (dense forward that is followed by a compacted backward): row_src, dst_of, total
int launch_gather_index(const int* keep_local, const int* counts, int* total, const float* maskf, int* row_src, int* dst_of,
                        int batch, hipStream_t s, int* drop_src = nullptr);   // drop_src: + the dropped tokens' rows, ascending (total[1] of them)

// last block: LayerNorm of the cls rows only (out[b] = LN(u[b*197])), stats[b*197], u_cls[b] = AT(u[b*197])
int launch_ln_cls(int precision, const float* u, const float* w, const float* b, void* out, float2* stats, void* u_cls,
                  int batch, hipStream_t s);
int launch_cls_index(int* cls_rows, int batch, hipStream_t s);

int launch_im2col(int precision, const float* images, void* out, int batch, hipStream_t s);
int launch_cls_rows(const float* cls, const float* pos, float* x0, int batch, hipStream_t s);
// fp32 -> AT copy (n elements)
int launch_convert(int precision, const float* src, void* dst, int64_t n, hipStream_t s);
// dst[c][r] = src[r][c]  (rows x cols fp32 -> AT, transposed), optional zero padding of dst rows/cols
int launch_transpose_convert(int precision, const float* src, void* dst, int rows, int cols, int dst_rows,
                             int dst_cols, hipStream_t s);
// dst[dst_rows][dst_cols] (zero padded) = src[rows][cols]
int launch_pad_convert(int precision, const float* src, void* dst, int rows, int cols, int dst_rows,
                       int dst_cols, hipStream_t s);

// head: final LayerNorm on the cls rows + Linear(768, C)
int launch_head_fwd(const float* x, const float* nw, const float* nb, const float* hw, const float* hb,
                    float* cls_n, float2* stats, float* logits, int batch, int C, hipStream_t s);
// g (all rows) = 0 except cls rows = LNbwd(dlogits @ Wh); dWh += dlogits^T cls_n ; dbh += colsum(dlogits)
int launch_head_bwd(const float* dlogits, const float* x, const float* cls_n, const float2* stats,
                    const float* nw, const float* hw, float* g, float* dWh, float* dbh, int batch, int C,
                    int compact, hipStream_t s);  // compact: g is [B,768] (cls rows only), no zero fill

struct LossArgs {
    const float* logits_s; const float* logits_t; const int64_t* targets;
    const int* counts;      // [depth*B] kept tokens per (layer,image) incl. cls, student pass
    int batch, C, depth;
    int count_batch = 0;    // images the kept-token counts span (video: frames = batch * t); 0 = batch
    float target_ratio, loss_ratio, token_minimal, token_minimal_weight;
    float* dlogits_s; float* dlogits_t; float* out_losses; float* dtok;
    float* scratch = nullptr;   // [4*B] per-image partial terms
    // class-probability targets [B, C] (timm Mixup's output through the reference's engine_finetune.py:44-45 into nn.CrossEntropyLoss): CE = -sum_c t_c log p_c
    // for both passes, d logits = p sum(t) - t; null = the integer labels in `targets`
    const float* soft = nullptr;
};
int launch_loss(const LossArgs& a, hipStream_t s);

// seed word on the device: set / advance by one (one thread)
int launch_seed_set(uint64_t* seed_dev, uint64_t value, hipStream_t s);
int launch_seed_advance(uint64_t* seed_dev, hipStream_t s);
// global-norm gradient clipping (torch.nn.utils.clip_grad_norm_ on the flat buffer): norm_out[0] = ||pre_scale * g||_2,
// then g *= min(1, max_norm / (norm + 1e-6)); scratch: 256 floats
int launch_clip_grad_norm(float* g, int64_t n, float max_norm, float pre_scale, float* scratch, float* norm_out, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                 float eps, float wd, float bc1, float bc2, float gscale, hipStream_t s);

// ... guarded: skipped (and counted) when the gradient holds inf / NaN; state = device int[4] {applied, skipped, flag, -}; bias corrections
// from the device-side applied count
int launch_adamw_guarded(float* p, const float* g, float* m, float* v, int64_t n, int* state, float lr, float b1, float b2,
                         float eps, float wd, float gscale, hipStream_t s);

// backward prep for one block: g_at = AT(g) ; dH[dst_of[t]] = AT(g[t] * mask) ;
// dmask[token] = <g[token], h[r]> for kept rows (0 elsewhere)
struct BwdPrepArgs {
    const float* g;         // [M,768]
    const void* h;          // [K,768] AT saved MLP output (null: no gate gradient, e.g. teacher pass)
    const int* dst_of;      // [M] token -> compact row (-1 = dropped), or null (dense: identity)
    const float* row_mask;  // masked-dense mode: per token mask (dH = mask*g), else null
    void* g_at;             // [M,768] AT (null when AT == float: g itself is used)
    void* dH;               // [K,768] AT (null when dense and unmasked: g_at / g is used)
    float* dmask;           // [M] (zero-filled by the kernel for rows it does not own)
    int M;
    float gs = 1.0f;        // factor the gradient carries wherever it is held in the 16-bit operand type (fp16 build: 2^12)
};
int launch_bwd_prep(int precision, const BwdPrepArgs& a, hipStream_t s);

// per-token tail of a block's backward:
//   du[t] = du[t] + LN2bwd(dA2[dst_of[t]]) (kept tokens) + dlogit[t]*wg ; du_at = AT(du)
//   dlogit = (dmask[t] + dm_ext) * s(1-s)/tau + dtoken_logits ; partial dwg / dbg per block of rows
struct TokBwdArgs {
    float* du;                 // [M,768] in/out (holds g (+ the adapter dgrad when `dad` is null) on entry)
    const void* dA2;           // [K,768] AT gradient w.r.t. LN2 output (null: skip MLP part)
    const int* dst_of;         // [M] compact row of a token or -1 ; null = identity (dense)
    const float* u;            // [M,768]
    const float2* stats2;      // [M]
    const float* ln2_w;
    const float* gate_w;       // [768] (null: no gate backward, e.g. teacher pass)
    const float* soft;         // [M] y_soft
    const float* maskf;        // [M]
    const float* dmask;        // [M] <g,h>
    const float* dtoken_select;  // user gradient (pointer offset to this layer, stride out_stride) or null
    const float* dtoken_logits;  // idem or null
    const float* dtok;           // device float[3] uniform terms or null
    int out_stride;
    int training;              // eval: y_soft = sigmoid(logit), same derivative form with tau = 1
    float tau;
    void* du_at;               // AT copy of du (null when AT == float or not needed)
    float* partial;            // [nblocks][769] dwg / dbg partials
    int M;
    int write_du;              // 0 for block 0 (du itself is not needed)
    float gs = 1.0f, inv_gs = 1.0f;   // 16-bit gradient operands (dA2, dad in; du_at out) carry the factor gs (fp16 build)
    const void* dad = nullptr;     // [M,768] AT adapter dgrad to add to du (null: already accumulated into du)
    const float* g_cls = nullptr;  // last block: incoming gradient exists for the cls rows only ([B,768]);
                                   // du/dA2 are then read as (n == 0 ? g_cls[b] / dA2[b] : 0)
    // The saved MLP output of a kept token includes the adapter's weight term (h' = mlp(x) + s d_act W_up^T: the fc2 GEMM carried
    // the up-projection as its leading k-tile; the fc2 epilogue leaves s b_up out of what it saves), so dmask = <g, h'> is too
    // large by <g, s d_act W_up^T>.  With ddz = the up-projection dgrad (g s W_up masked by relu' / dropout, times inv_keep and
    // gs) that is <d_act, ddz> / (inv_keep gs): subtracted here, per kept token.  cat_dact == null: h is the MLP output alone.
    const void* cat_dact = nullptr;   // [M, 64] AT (token-indexed)
    const void* cat_ddz = nullptr;    // [M, 64] AT
    const float* cat_bup = nullptr;   // unused (kept for ABI stability of the struct)
    float cat_scale = 0.f, cat_ddz_scale = 0.f;   // s ; 1 / (inv_keep * gs)
    void* du3 = nullptr;           // fp32 split form: + du * du3_scale as the [M][3*768] 16-bit hi / hi / lo operand of the proj dgrad
    float du3_scale = 1.0f; bool du3_hi_only = false;   // hi_only: without the lo half (one-part proj dgrad)
    // stochastic depth: the MLP branch of image b was multiplied by branch_scale[b] in the forward pass (GemmArgs::row_scale of FC2), so
    // its LN2-input gradient and the gate gradient <g, h> are too
    const float* branch_scale = nullptr;
    // fp16 mode (round 6, dyt_ctx::g16): the gradient stream between the row kernels of the backward lives in ONE 16-bit, gs-scaled buffer per
    // hop -- the copy each kernel writes for the GEMM behind it anyway (ln_bwd's g_at, this kernel's du_at) -- instead of also as an fp32
    // [M,768] stream: du_in_at = the incoming gradient in that form (du is then not read), du == null: no fp32 copy is written
    const void* du_in_at = nullptr;
    // (u from the forward's 16-bit copy instead of the fp32 row was measured too: 23.23 / 23.26 vs 23.23 / 23.10 ms per step same-box -- no gain, not kept:
    // profiles/round6/r6_u16_ab.txt)
};
int launch_tok_bwd(int precision, const TokBwdArgs& a, int* nblocks_out, hipStream_t s);
// stochastic depth (timm DropPath): scales[branch][l][b] for branch 0 (attention) / 1 (MLP), blocks l < depth, images b < batch: 1 with
// probability keep_l = 1 - rate * l / (depth - 1), else 0, divided by keep_l; block 0 (rate 0) always 1.  Philox stream (seed | *seed_dev, subseq_base + 2 l + branch, b)
int launch_drop_path_draw(float* scales, int depth, int batch, float rate, uint64_t seed, const uint64_t* seed_dev, uint64_t subseq_base, hipStream_t s);
// x[row, :] *= scale[row / 197] for an AT matrix [M, ld] (the attention branch's gradient after the proj dgrad)
int launch_scale_rows(int precision, void* x, const float* scale, int M, int ld, hipStream_t s);
// out[i] += alpha * sum_p partial[p*stride + i], i < n
int launch_reduce_partials(const float* partial, int nparts, int stride, float* out, int n, float alpha,
                           hipStream_t s);

// wgrad: C[c][j] = sum_m X[m][c] * Y[m][j] (X [M,768] AT, Y [M,64] AT) via per-chunk partials;
//   out_w[c*sc + j*sj] += alpha * C[c][j]  (j < r) ;  out_xsum[c] += alpha_x * sum_m X[m][c] (optional)
struct WgradArgs {
    const void* X; const void* Y; int M; int r;
    float* partial;            // scratch [nchunks][768 + 8][80]
    float* out_w; int sc, sj; float alpha;
    float* out_xsum; float alpha_x;
    float* out_ysum = nullptr; float alpha_y = 0.f;   // out_ysum[j] += alpha_y * sum_m Y[m][j], j < r
    // fp32 mode only: the products on the 16-bit matrix cores from fp32 inputs converted as they are loaded (one-part gradient
    // products of "fp16x3f"); X / Y are multiplied by x_scale / y_scale (powers of two) first, the outputs divided again
    bool half_products = false; float x_scale = 1.f, y_scale = 1.f;
};
// Reductions of per-chunk partials deferred to ONE batched launch (the backward pass queues the adapter weight-gradient and the
// gate-gradient reductions of several blocks -- each with its own partial buffer -- and flushes them where the gradients have to
// be final: 36 latency-bound launches per pass become 4).  Fixed summation order per output, as the immediate form.
struct WgReduceDesc {
    const float* partial; float* out_w; int sc, sj; float alpha; float* out_xsum; float alpha_x; float* out_ysum; float alpha_y;
    int nchunks;
};
struct ReduceQueue {
    static constexpr int MAX_WG = 32, MAX_TOK = 16;
    WgReduceDesc wg[MAX_WG]; int n_wg = 0; int r = 0;
    const float* tok_partial[MAX_TOK]; float* tok_out[MAX_TOK]; int tok_nparts[MAX_TOK]; int n_tok = 0;
};
int launch_wgrad(int precision, const WgradArgs& a, hipStream_t s);
// two products with the same M and r (separate `partial` buffers) as one launch + one reduce launch (defer: queued instead)
int launch_wgrad(int precision, const WgradArgs* a, int n, hipStream_t s, ReduceQueue* defer = nullptr);
// queue out[i] += sum_p partial[p * 769 + i], i < 769 (the gate weight + bias gradient partials of tok_bwd)
int queue_tok_reduce(ReduceQueue& q, const float* partial, int nparts, float* out);
int flush_reductions(ReduceQueue& q, hipStream_t s);
// ------------------------------------------------------------------------------------------
// video model: attentive pooling head (pool.hip)
// ------------------------------------------------------------------------------------------
int launch_pool_ln_fwd(int precision, const float* x, const float* nw, const float* nb, const float* kw, const float* kb,
                       const float* vw, const float* vb, float* xf, float2* st_f, float2* st_kv, void* xk, void* xv,
                       int rows, hipStream_t s);
int launch_pool_q_fwd(const float* query, const float* nqw, const float* nqb, const float* Wq, const float* qbias, float* qn,
                      float* qhat, float* st_q, float* qs, hipStream_t s);
int launch_pool_attn_fwd(int precision, const float* qs, const void* K, const void* V, float* P, float* o, int clips,
                         int NK, hipStream_t s);
int launch_pool_attn_bwd(int precision, const float* qs, const void* K, const void* V, const float* P, const float* dO,
                         void* dK, void* dV, float* dq_part, int clips, int NK, float gs, hipStream_t s);
int launch_pool_q_bwd(const float* dq_part, int clips, const float* qn, const float* qhat, const float* st_q, const float* Wq,
                      const float* nqw, float* gq, float* dqn, float* dWq, float* dqb, float* dnqw, float* dnqb,
                      float* dquery, hipStream_t s);
int launch_rows_linear(const float* x, const float* W, const float* bias, float* out, int R, int N, int K, float scale,
                       hipStream_t s);
int launch_rows_linear_bwd(const float* dout, const float* x, const float* W, float* dx, float* dW, float* db, int R, int N,
                           int K, hipStream_t s);
int launch_transpose_rows(int precision, const void* src, void* dst, int rows, int rows_pad, float* colsum_part,
                          hipStream_t s);
int launch_pool_ln_bwd(int precision, const void* dxk, const void* dxv, const float* xf, const float2* st_kv, const float* kw,
                       const float* vw, const float* x, const float2* st_f, const float* nw, float* g, float* partial,
                       int rows, int* nblocks_out, float gs, hipStream_t s);

inline size_t at_size(int precision) { return precision == 0 ? 4 : 2; }

// Measurement hook (tools/probes/marginal_cost.sh): DYT_DBG_SKIP = bit mask of kernel classes whose launches are dropped (results
// are garbage; the step time then shows what that class costs in the overlapped schedule, where serial durations do not add up)
//   1 attention bwd  2 attention fwd  4 adapter weight gradients  8 tok_bwd  16 ln_bwd  32 ln_fwd  64 GEMMs with K = 64 or N = 64 (adapter)
//   128 N = 768 GEMMs (K >= 256)  256 wide GEMMs (N >= 2304)
//   1024: the mask applies to launches enqueued by the backward pass only (the forward pass, hence the kept-token counts, stays intact)
extern int g_dbg_in_backward;
#ifdef DYT_DEBUG_HOOKS   // measurement builds only (make CXXFLAGS_EXTRA=-DDYT_DEBUG_HOOKS; a command-line CXXFLAGS= would replace the Makefile's own flags): a product build cannot be talked into dropping launches by an environment variable
inline bool dbg_skip(int bit) {
    static int mask = -1;
    if (mask < 0) { const char* e = getenv("DYT_DBG_SKIP"); mask = e ? atoi(e) : 0; }
    if ((mask & 1024) && !g_dbg_in_backward) return false;
    return (mask & bit) != 0;
}
#else
inline bool dbg_skip(int) { return false; }
#endif

}  // namespace dyt
