// libdyt_hip.so: context, whole-model forward / backward orchestration and the C ABI
// declared in include/dyt_hip.h.  Host side only enqueues kernels on the caller's stream: no
// host<->device synchronisation, no allocation after dyt_ctx_create, kept-token counts stay on
// the device (GEMM grids are sized for the dense case and tiles beyond the device-side row count
// exit immediately), so a step is graph-capturable.
//
// Mirrors (paths relative to the reference root):
//   VisionTransformer.forward_features/forward_head  models/vision_transformer_IN21K.py:343-385
//   Block.forward                                    models/vision_transformer_IN21K.py:144-165
//   compacted MLP                                    models/model_speed_test.py:274-310
//   step body                                        engine_finetune.py:47-79
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/dyt_hip.h"
#include "kernels.h"

namespace dyt {

int g_dbg_in_backward = 0;   // measurement hook, see dbg_skip (kernels.h)
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------
// small kernels private to this file
// ------------------------------------------------------------------------------------------
// Per-step refresh of the adapter weights in the layouts / dtype the GEMMs want:
//   down_w  [RP,768] (rows >= r zero)      forward down-projection (N = RP, K = 768)
//   down_wT [768,RP]                       dgrad through down_proj (N = 768, K = RP)
//   up_w    [768,RP] (cols >= r zero)      forward up-projection   (N = 768, K = RP)
//   up_wT   [RP,768]                       dgrad through up_proj   (N = RP, K = 768)
//   down_b  [RP] fp32
template <class AT>
__global__ void prep_adapters_kernel(const float* __restrict__ flat, int64_t layer_stride, int64_t off_dw, int64_t off_db,
                                     int64_t off_uw, int r, AT* __restrict__ down_w, AT* __restrict__ down_wT,
                                     AT* __restrict__ up_w, AT* __restrict__ up_wT, float* __restrict__ down_b,
                                     AT* __restrict__ up_ws, float scale, int64_t off_sc, int64_t off_ub, float* __restrict__ up_bp, float bias_scale) {
    // off_sc >= 0 ("learnable_scalar", DYT_OPT_LEARNABLE_SCALE): the up-projection copies and up_bp [depth][768] carry the block's trainable
    // scale s = flat[off_sc] (W' = s W_up, b' = s b_up), and every kernel downstream runs with scale 1
    const int l = blockIdx.y;
    const float* base = flat + (int64_t)l * layer_stride;
    const float ls = off_sc >= 0 ? base[off_sc] : 1.0f;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    constexpr int SZ = RP * D;
    if (idx >= SZ) return;
    {   // idx -> (j, c) of [RP,768]
        const int j = idx / D, c = idx - j * D;
        const float dw = j < r ? base[off_dw + (int64_t)j * D + c] : 0.f;   // down_proj.weight [r,768]
        const float uw = j < r ? ls * base[off_uw + (int64_t)c * r + j] : 0.f;   // up_proj.weight [768,r]
        down_w[(size_t)l * SZ + idx] = from_f32<AT>(dw);
        up_wT[(size_t)l * SZ + idx] = from_f32<AT>(uw);
    }
    {   // idx -> (c, j) of [768,RP]
        const int c = idx / RP, j = idx - c * RP;
        const float dw = j < r ? base[off_dw + (int64_t)j * D + c] : 0.f;
        const float uw = j < r ? ls * base[off_uw + (int64_t)c * r + j] : 0.f;
        down_wT[(size_t)l * SZ + idx] = from_f32<AT>(dw);
        up_w[(size_t)l * SZ + idx] = from_f32<AT>(uw);
        if (up_ws) up_ws[(size_t)l * SZ + idx] = from_f32<AT>(scale * uw);
    }
    if (idx < RP) down_b[l * RP + idx] = idx < r ? base[off_db + idx] : 0.f;
    if (up_bp && idx < D) up_bp[l * D + idx] = bias_scale * ls * base[off_ub + idx];   // bias_scale: the adapter's LayerNorm "out" form takes s b_up here
}

// "learnable_scalar": the backward has left G' = dL/dW', gb' = dL/db' of the PRIMED up-projection (W' = s W_up, b' = s b_up) of every block
// in scr (the slot's scratch, flat layout); chain rule into the gradient buffer: dW_up += s G', db_up += s gb', ds += <G', W_up> + <gb', b_up>.
// One workgroup per block, fixed reduction order.
__global__ __launch_bounds__(256) void learn_scale_fixup_kernel(const float* __restrict__ scr, const float* __restrict__ flat, float* __restrict__ grad,
                                                                int64_t layer_stride, int64_t off_uw, int64_t off_ub, int64_t off_sc, int r, int l0) {
    __shared__ float red[256];
    const int l = l0 + blockIdx.x, tid = threadIdx.x;
    const int64_t b = (int64_t)l * layer_stride;
    const float ls = flat[b + off_sc];
    float dot = 0.f;
    for (int i = tid; i < D * r; i += 256) {
        const float g = scr[b + off_uw + i];
        dot = fmaf(g, flat[b + off_uw + i], dot);
        grad[b + off_uw + i] += ls * g;
    }
    for (int i = tid; i < D; i += 256) {
        const float g = scr[b + off_ub + i];
        dot = fmaf(g, flat[b + off_ub + i], dot);
        grad[b + off_ub + i] += ls * g;
    }
    red[tid] = dot;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) grad[b + off_sc] += red[0];
}

template <class AT>
__global__ void qkv_split_kernel(const float* __restrict__ qkv, AT* __restrict__ q, AT* __restrict__ k, AT* __restrict__ v,
                                 int batch) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)batch * NT * 3 * D) return;
    const int col = (int)(idx % (3 * D));
    const size_t row = idx / (3 * D);
    const int b = (int)(row / NT), n = (int)(row % NT);
    const int which = col / D, c = col % D, h = c >> 6, d = c & 63;
    const float val = qkv[idx] * (which == 0 ? 0.125f : 1.0f);
    AT* dst = which == 0 ? q : (which == 1 ? k : v);
    dst[(((size_t)b * NH + h) * NT + n) * HD + d] = from_f32<AT>(val);
}
template <class AT>
__global__ void to_f32_kernel(const AT* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = to_f32(src[i]);
}

}  // namespace dyt

using namespace dyt;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct LayerW {  // frozen, library-owned
    float *ln1_w, *ln1_b, *qkv_b, *proj_b, *ln2_w, *ln2_b, *fc1_b, *fc2_b;
    void *qkv_w, *qkv_wT, *proj_w, *proj_wT, *fc1_w, *fc1_wT, *fc2_w, *fc2_wT;
    void *qkv_wp = nullptr, *fc1_wp = nullptr, *fc2_wTp = nullptr;   // bf16 mode: MFMA-fragment-order twins (gemm_bpre.h)
    void *proj_wp = nullptr, *proj_wTp = nullptr, *qkv_wTp = nullptr, *fc1_wTp = nullptr;
    // fp32 mode: [N, 2K] 16-bit [hi | lo] images of the eight matrices (DYT_OPT_F32_SPLIT16, launch_split3_w)
    void *qkv_w3 = nullptr, *qkv_wT3 = nullptr, *proj_w3 = nullptr, *proj_wT3 = nullptr, *fc1_w3 = nullptr, *fc1_wT3 = nullptr, *fc2_w3 = nullptr, *fc2_wT3 = nullptr;
    // fp32 mode with a 16-bit backward ("fp16x3h", dyt_ctx::bwd16): the transposed matrices the dgrad GEMMs multiply by, in the 16-bit
    // operand type, plain [in,out] and in MFMA fragment order (what the 16-bit mode keeps as *_wT / *_wTp)
    void *fc1_w3b = nullptr, *fc2_w3b = nullptr, *qkv_w3b = nullptr, *proj_w3b = nullptr;   // second image of a class whose two passes take different forms (student three-part, teacher fp8-correction)
    int* w_exp = nullptr;   // "fp16f8": device words with the exponents of the four forward images (qkv, proj, fc1, fc2), launch_split_w_f8
    void *qkv_wT16 = nullptr, *qkv_wTp16 = nullptr, *proj_wT16 = nullptr, *proj_wTp16 = nullptr, *fc1_wT16 = nullptr, *fc1_wTp16 = nullptr,
         *fc2_wT16 = nullptr, *fc2_wTp16 = nullptr;
    // 16-bit modes, LayerNorm-2 folded into fc1 (dyt_ctx::ln_fold): AT(gamma W1) plain + in fragment order, its column sums, b1 + W1 beta
    void *fc1_wf = nullptr, *fc1_wfp = nullptr; float *fc1_cs = nullptr, *fc1_bf = nullptr, *fc1_w32 = nullptr;   // (w32: the fp32 source, so that gamma W is rounded once)
};
struct LayerS {  // saved activations of one pass
    float2 *st1, *st2;
    void *q, *k, *v, *attn_o, *u_at, *z, *d_act;
    float *lse, *u, *soft, *maskf;
    void* h;
    int *keep_local, *offsets, *total, *row_src, *dst_of;
    float2* ln_part = nullptr;    // dyt_ctx::ln_fold: the proj epilogue's per-row LayerNorm partials of u ([M][LN_PARTS])
    float2* st_a = nullptr;       // dyt_ctx::ad_ln: (mean, rstd) of the adapter's LayerNorm input, [M]
    float* up32 = nullptr;        // dyt_ctx::ad_ln == 2: the LayerNorm's input s (d_act W_up^T + b_up), fp32 [M,768] (its backward needs x_hat)
    bool h_has_adapter = false;   // h = mlp(x) + s up(d_act) + s b_up (fc2 carried the up-projection, DYT_OPT_FC2_CAT)
    // dyt_ctx::bwd16: what the backward pass reads, in the 16-bit operand type (written by the exact forward next to / instead of
    // the fp32 tensors above: q16 / k16 / v16 / o16 by the split attention kernel, u16 by the proj epilogue, z16 = gelu'(z) by the
    // fc1 epilogue, dact16 by the down-projection epilogue, h16 by the fc2 epilogue)
    void* ao3 = nullptr;   // bwd16: the attention output as the proj GEMM's split operand image, kept per layer: its hi plane (row stride SPLIT_A * 768) IS the 16-bit output the backward reads (no separate o16 write: 14 of the forward kernel's 125 us)
    void *q16 = nullptr, *k16 = nullptr, *v16 = nullptr, *o16 = nullptr, *u16 = nullptr, *z16 = nullptr, *dact16 = nullptr, *h16 = nullptr;
};
struct Transients {  // scratch of one pass (per slot, so two passes can run on two streams)
    void *xn, *h1, *g_at, *dZ, *ddz, *du_at, *dO, *dqkv, *dA2, *dxn, *dad;
    void* dact_s = nullptr;   // 16-bit modes: adapter_scale * d_act, the A2 operand of the fc2 + up-projection contraction
    void* dact3 = nullptr;    // 16-bit-backward split modes: the same as a [M][hi 64 | lo 64] image (three-part up-projection, alone or as the fc2 GEMM's leading tiles)
    void* a3 = nullptr;   // fp32 mode: [M, 3 * 3072] 16-bit scratch for the split A operand of a GEMM
    void* g3 = nullptr;   // [M, 3*768]: the gradient stream as a split operand, written by ln_bwd (next block's GELU' dgrad) and tok_bwd (proj dgrad); attention output in the forward pass
    void *xn3 = nullptr, *h3 = nullptr, *dqkv3 = nullptr;   // dqkv3: [M, 3*2304] from the attention backward   // ... and the split operands producers write directly: LN output [M, 3*768], fc1 output / dZ [M, 3*3072]
    float *g, *delta, *dmask, *tok_partial, *wg_partial, *wg_partial2;
    void *qlo = nullptr, *klo = nullptr, *vlo = nullptr;   // dyt_ctx::bwd16: lo planes of q / k / v (the hi planes are LayerS::q16 / k16 / v16), QKV epilogue -> split attention forward
    int* drop_src = nullptr;        // dyt_ctx::ln_fold: token rows of the DROPPED tokens of the block in flight (gather_index -> their up-projection launch)
    float2* st_compact = nullptr;   // dyt_ctx::ln_fold: scratch for LN2's (mean, rstd) in logical-row order (GemmArgs::ln_scratch)
    void* dad16 = nullptr;   // dyt_ctx::bwd16: the adapter dgrad as a 16-bit [M,768] operand of tok_bwd (T.dad of the 16-bit modes)
    // dyt_ctx::ad_ln: fp32-sized [M,768] scratch each -- xa: LayerNorm_a(u) in the operand type of the pass that reads it ("in"; recomputed by the backward);
    // dup: the gradient behind the adapter's LayerNorm ("out") / the fp32 adapter dgrad of an fp32 backward ("in"); ln_part: parameter-gradient partials
    float *xa = nullptr, *dup = nullptr, *aln_part = nullptr;
};
struct PoolS {  // video pooling head: saved activations of one pass (pool.hip)
    float *xf = nullptr, *P = nullptr, *o = nullptr, *y = nullptr, *qn = nullptr, *qhat = nullptr, *qs = nullptr, *st_q = nullptr;
    float *dq_part = nullptr, *gq = nullptr, *dqn = nullptr, *dy = nullptr, *dO = nullptr;
    float2 *st_f = nullptr, *st_kv = nullptr;
    void *xk = nullptr, *xv = nullptr, *Kp = nullptr, *Vp = nullptr;
    void *dKt = nullptr, *xkt = nullptr, *dVt = nullptr, *xvt = nullptr;   // [768, Mpad] token-transposed operands of the k / v weight gradients
    hipStream_t wstream = nullptr; hipEvent_t ev_wf = nullptr, ev_wj = nullptr; bool wpending = false;   // weight-gradient side stream
};
struct Slot {
    Transients T;
    PoolS pool;
    hipStream_t branch = nullptr;        // side stream for the adapter branch of this pass
    float* u0_own = nullptr; void* u0_at_own = nullptr;  // block-0 buffers (slot 1 may alias slot 0's, see step)
    hipEvent_t ev_f = nullptr, ev_j = nullptr;
    bool no_branch = false;              // this pass runs without its adapter side stream (see dyt_step_fwd_bwd)
    std::vector<LayerS> L;
    std::vector<float*> wg_part, wg_part2, tok_part;   // per-layer partial buffers: their reductions are batched (ReduceQueue)
    std::vector<float*> xs;  // depth+1 residual-stream snapshots
    int* counts = nullptr;   // [depth*B]
    void* ucls_at = nullptr; // last block: AT(u[cls rows]) [B,768] (adapter-down operand, kept for its wgrad)
    void* ucls16 = nullptr;  // dyt_ctx::bwd16: its 16-bit copy
    void* u0_16_own = nullptr;   // block-0 u16 of this slot (slot 1 may alias slot 0's, like u0_own)
    float2* part0_own = nullptr; // block-0 LayerNorm partials of this slot (idem)
    float* gcls = nullptr;   // last block backward: gradient at the cls rows [B,768]
    float* cls_n = nullptr;
    float2* head_stats = nullptr;
    int batch = 0, flags = 0;
    bool valid = false;
    bool saved16 = false;    // the saved pass holds the 16-bit tensors of a 16-bit backward (dyt_ctx::bwd16)
    const float* trainable = nullptr;  // flat trainable buffer the saved pass was computed with
    // stochastic depth (dyt_set_drop_path): the per-image branch factors [2][depth][batch] of this slot's passes -- drawn by the library
    // into dp_own at every training forward, or the caller's (dp_inject, dyt_set_drop_path_scales); dp = what the SAVED pass used (null: none)
    float* dp_own = nullptr; const float* dp_inject = nullptr; const float* dp = nullptr;
    float* gscr = nullptr;   // DYT_OPT_LEARNABLE_SCALE: [depth * layer_stride] the backward's up-projection gradients before the chain rule through the scale
};
struct ProfRec { int cat; double flops; hipEvent_t a, b; const int* m_dev; int M; };

struct dyt_ctx {
    dyt_config cfg;
    int prec;
    size_t at;  // bytes per activation element
    char* arena = nullptr;
    size_t arena_size = 0, arena_used = 0;
    // second arena, allocated when DYT_OPT_F32_SPLIT16 is first switched on: everything only the split forms of the fp32 mode use
    // (the [hi | lo] weight images, the split operand scratch, the 16-bit tensors of a 16-bit backward) -- a plain fp32 context
    // does not carry it
    char* aux_arena = nullptr;
    size_t aux_size = 0;
    bool aux_bwd16 = false;     // the aux arena holds the bwd16 buffers
    int f8_mask_complete = 0;   // ... of a complete_model (teacher) pass: no token-keep decision depends on it (its gate output is discarded), only its logits -- "fp16x3q": 15 + 32 (32 = its attention forward as the hi * hi product alone)
    int f8_mask = 0;            // classes of forward GEMMs in that form: 1 qkv, 2 proj, 4 fc1, 8 fc2, 16 patch embedding (DYT_F8_CLASSES; "fp16f8" = 31, "fp16x3q" = 3)
    bool f8 = false;            // "fp16f8": forward GEMMs as hi * hi on the f16 matrix cores + the two correction products on the fp8 ones (DYT_OPT_F32_SPLIT16 = 4; implies bwd16)
    int* pe_w_exp = nullptr; unsigned* f8_scratch = nullptr;
    bool bwd16 = false;         // "fp16x3h": the fp16x3 forward, the backward on the 16-bit mode's operands and kernels (DYT_OPT_F32_SPLIT16 = 3)
    void *ad_up_wT16 = nullptr, *ad_down_wT16 = nullptr, *ad_scratch16 = nullptr;   // bwd16: per-step 16-bit copies of the adapter matrices the dgrads read
    void* ad_up_w3 = nullptr;   // bwd16: [depth][768][hi 64 | lo 64] image of the (fp32) up-projection copies: the three-part up-projection's weight operand
    // frozen
    float *cls, *pos, *pe_b, *norm_w, *norm_b;
    void* pe_w;
    std::vector<LayerW> W;
    // per-step AT copies of the adapters (all layers contiguous)
    void *ad_down_w, *ad_down_wT, *ad_up_w, *ad_up_wT;
    float* ad_down_b;
    bool split_attn = true;     // ... and the attention forward too (attn_fwd_split_kernel; DYT_SPLIT_ATTN=0: exact fp32 MFMA kernel)
    bool split_prod = true;     // ... attention forward / ln_bwd / tok_bwd write the split operand of the GEMM that follows (DYT_SPLIT_PROD=0: pre-passes)
    float split_gs = 4096.0f;   // ... their gradient operands are multiplied by this power of two before the split (DYT_SPLIT_GS_LOG2)
    int split_bwd_parts = 3;    // ... products of the GRADIENT GEMMs' contraction (DYT_SPLIT_BWD_PARTS: 3 full, 2 = dY_hi * (W_hi + W_lo), 1 = dY_hi * W_hi)
    int split_bwd_attn_parts = 3;   // ... and of the split attention backward's dP / dQ / dK / dV products (3 or 1; the score recomputation keeps three)
    const float* soft_targets = nullptr; int soft_batch = 0;   // dyt_set_soft_targets: class-probability targets of the next loss evaluations
    int one_part_complete = 0;  // ... classes of a complete_model (teacher) pass contracted as hi * hi alone (SPLIT_F)
    int split_fwd_parts[4] = {3, 3, 3, 3};   // ... products of the FORWARD GEMMs per class (qkv, proj, fc1, fc2): measurement knob
    bool split_wgrad16 = true;  // ... adapter weight gradients as one-part products too (DYT_SPLIT_WGRAD16=0: the exact-fp32 kernel)
    bool split16 = false;       // fp32 mode: frozen-weight GEMMs as three 16-bit MFMA products (DYT_OPT_F32_SPLIT16)
    void* pe_w3 = nullptr;
    // 16-bit modes: LayerNorm-2 is not a kernel -- the proj epilogue emits per-row partial statistics of u, fc1 contracts the 16-bit copy
    // of u with gamma-folded weights and normalises in its epilogue (GemmArgs::ln_part).  -0.3 ms of a 26.8 ms step; vs the
    // oracle at B=16 as accurate as the kernel form (logits 1.6e-3 vs 1.8e-3).  Environment DYT_LN_FOLD=0 when the context is created:
    // the ln_fwd / ln_gather kernels (the reference's autocast order: LayerNorm in fp32, then round)
    bool ln_fold = true;
    bool fc2_cat = true;        // 16-bit modes: adapter up-projection rides on the fc2 GEMM where no separate h is needed
    // trainable flat layout
    int64_t layer_stride, off_dw, off_db, off_uw, off_ub, off_gw, off_gb, off_hw, off_hb, n_train;
    // video model (frames > 1): attentive pooling head, trainable
    int frames = 1;
    int64_t off_pquery = 0, off_pnq_w = 0, off_pnq_b = 0, off_pnk_w = 0, off_pnk_b = 0, off_pnv_w = 0, off_pnv_b = 0,
            off_pq_w = 0, off_pk_w = 0, off_pv_w = 0, off_pq_bias = 0, off_pv_bias = 0, off_pproj_w = 0, off_pproj_b = 0;
    void *pk_w = nullptr, *pk_wT = nullptr, *pv_w = nullptr, *pv_wT = nullptr;   // per-step AT copies of k / v weights
    std::vector<Slot> slots;
    float *dl_s, *dl_t, *dtok, *logits_s, *logits_t, *losses, *grad2, *loss_part;
    int* cls_rows = nullptr;   // [max_batch] token row of each image's cls token (b*197)
    uint64_t* seed_dev = nullptr;   // device-side Philox seed word (DYT_F_DEVICE_SEED: captured graphs draw fresh noise per replay)
    float* clip_scratch = nullptr;  // [256] partial sums of dyt_clip_grad_norm
    // gradient availability for a chunked all-reduce (dyt_stream_wait_grads): layers >= grad_split and the head are final
    hipEvent_t ev_half_s = nullptr, ev_half_t = nullptr, ev_upper = nullptr, ev_comm = nullptr;
    hipStream_t aux = nullptr;       // sums the upper part of the two passes' gradient buffers while the backward goes on
    bool upper_recorded = false;
    float gs = 1.0f;           // factor the gradient carries wherever it is held in the 16-bit operand type (fp16 build: 2^12, a
                               // fixed loss scale confined to the library: fp32 streams and every returned gradient are unscaled)
    bool cls_tail = true;      // last block: MLP/adapter on the cls rows only (only they reach the head)
    // second stream: the student and the teacher pass of a step are independent and run concurrently
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_b0 = nullptr;
    bool overlap = true;                 // any stream overlap
    bool ov_pass = true, ov_branch = false;  // student / teacher passes on two streams; adapter branch on its own stream
    bool ov_bwd_serial = false;          // the teacher's backward starts after the student's (bit-reproducible schedule)
    bool share_block0 = true;  // step: the teacher pass reuses the student's embedding + block-0 attention branch
    bool learn_scale = false;    // DYT_OPT_LEARNABLE_SCALE: tuning_config.ffn_adapter_scalar == "learnable_scalar" (dynamic_adapter.py:101-102): the scale is
                                 // the trainable word off_sc of every block (a padding slot of the flat layout behind the gate bias)
    int64_t off_sc = 0;
    // fp16 build, 16-bit mode (round 6): between the row kernels of the backward the gradient stream is carried by the 16-bit gs-scaled copy each
    // of them writes for the GEMM behind it (ln_bwd: g_at, tok_bwd: du_at) instead of ALSO as an fp32 [M,768] stream that both read and
    // write: -230 MB per block and pass (tok_bwd / ln_bwd 336 -> 221 MB each).  Every hop rounds the stream to 11 bits (at 2^12 x its
    // value), so round-off accumulates over the 23 hops of a pass (fp16 mode, five seeds: up_proj 1.1e-3 -> 1.3e-3, head unchanged): not used by the
    // bfloat16 build (8 bits).  DYT_G16=0: the fp32 stream.
    bool g16 = false;
    // dyt_config::adapter_ln (tuning_config.ffn_adapter_layernorm_option): 0 none, 1 "in", 2 "out"; gamma / beta of block l at l * layer_stride + off_alw / off_alb
    int ad_ln = 0;
    int64_t off_alw = 0, off_alb = 0;
    void* ad_up_ws = nullptr;   // "out": [depth][768][RP] adapter_scale * up_proj.weight in the forward's operand type (the LayerNorm input is s (d_act W^T + b))
    bool pass_ran = false;       // a forward pass has run in this context: DYT_OPT_LEARNABLE_SCALE may no longer change
    float* ad_up_bp = nullptr;   // [depth][768] s * up_proj.bias (prep_adapters_kernel)
    float drop_path_rate = 0.f;  // timm DropPath rate of the LAST block (block l: rate * l / (depth - 1)); training forward passes only
    int count_flops_tokens = 0;  // > 0: Block.forward_count_flops -- MLP on the first n tokens of every image
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <class T>
static T* carve(dyt_ctx* c, size_t count, bool dry) {
    const size_t bytes = align_up(count * sizeof(T), 256);
    T* p = dry ? nullptr : reinterpret_cast<T*>(c->arena + c->arena_used);
    c->arena_used += bytes;
    return p;
}
static void* carve_at(dyt_ctx* c, size_t count, bool dry) {
    const size_t bytes = align_up(count * c->at, 256);
    void* p = dry ? nullptr : c->arena + c->arena_used;
    c->arena_used += bytes;
    return p;
}

static void layout(dyt_ctx* c, bool dry) {
    const dyt_config& cf = c->cfg;
    const size_t B = cf.max_batch, M = B * NT, depth = cf.depth, C = cf.num_classes;
    c->arena_used = 0;
    c->cls = carve<float>(c, D, dry);
    c->pos = carve<float>(c, NT * D, dry);
    c->pe_b = carve<float>(c, D, dry);
    c->norm_w = carve<float>(c, D, dry);
    c->norm_b = carve<float>(c, D, dry);
    c->pe_w = carve_at(c, (size_t)D * D, dry);
    c->W.resize(depth);
    for (size_t l = 0; l < depth; ++l) {
        LayerW& w = c->W[l];
        w.ln1_w = carve<float>(c, D, dry); w.ln1_b = carve<float>(c, D, dry);
        w.ln2_w = carve<float>(c, D, dry); w.ln2_b = carve<float>(c, D, dry);
        w.qkv_b = carve<float>(c, 3 * D, dry); w.proj_b = carve<float>(c, D, dry);
        w.fc1_b = carve<float>(c, DM, dry); w.fc2_b = carve<float>(c, D, dry);
        w.qkv_w = carve_at(c, (size_t)3 * D * D, dry); w.qkv_wT = carve_at(c, (size_t)3 * D * D, dry);
        w.proj_w = carve_at(c, (size_t)D * D, dry); w.proj_wT = carve_at(c, (size_t)D * D, dry);
        w.fc1_w = carve_at(c, (size_t)DM * D, dry); w.fc1_wT = carve_at(c, (size_t)DM * D, dry);
        w.fc2_w = carve_at(c, (size_t)DM * D, dry); w.fc2_wT = carve_at(c, (size_t)DM * D, dry);
        if (c->prec != 0) {
            w.qkv_wp = carve_at(c, (size_t)3 * D * D, dry); w.fc1_wp = carve_at(c, (size_t)DM * D, dry);
            w.fc2_wTp = carve_at(c, (size_t)DM * D, dry);
            w.proj_wp = carve_at(c, (size_t)D * D, dry); w.proj_wTp = carve_at(c, (size_t)D * D, dry);
            w.qkv_wTp = carve_at(c, (size_t)3 * D * D, dry); w.fc1_wTp = carve_at(c, (size_t)DM * D, dry);
            if (c->ln_fold) {
                w.fc1_wf = carve_at(c, (size_t)DM * D, dry); w.fc1_wfp = carve_at(c, (size_t)DM * D, dry);
                w.fc1_cs = carve<float>(c, DM, dry); w.fc1_bf = carve<float>(c, DM, dry); w.fc1_w32 = carve<float>(c, (size_t)DM * D, dry);
            }
        }
    }
    c->ad_down_w = carve_at(c, depth * RP * D, dry);
    c->ad_down_wT = carve_at(c, depth * RP * D, dry);
    c->ad_up_w = carve_at(c, depth * RP * D, dry);
    c->ad_up_wT = carve_at(c, depth * RP * D, dry);
    c->ad_down_b = carve<float>(c, depth * RP, dry);
    c->slots.resize(cf.slots);
    for (int s = 0; s < cf.slots; ++s) {
        Slot& S = c->slots[s];
        S.L.resize(depth);
        S.xs.resize(depth + 1);
        for (size_t l = 0; l <= depth; ++l) S.xs[l] = carve<float>(c, M * D, dry);
        S.counts = carve<int>(c, depth * B, dry);
        S.ucls_at = carve_at(c, B * D, dry);
        S.gcls = carve<float>(c, B * D, dry);
        S.cls_n = carve<float>(c, B * D, dry);
        S.head_stats = carve<float2>(c, B, dry);
        for (size_t l = 0; l < depth; ++l) {
            LayerS& L = S.L[l];
            L.st1 = carve<float2>(c, M, dry); L.st2 = carve<float2>(c, M, dry);
            if (c->ad_ln) L.st_a = carve<float2>(c, M, dry);
            if (c->ad_ln == 2) L.up32 = carve<float>(c, M * D, dry);
            L.q = carve_at(c, M * D, dry); L.k = carve_at(c, M * D, dry); L.v = carve_at(c, M * D, dry);
            L.attn_o = carve_at(c, M * D, dry);
            L.lse = carve<float>(c, B * NH * NT, dry);
            L.u = carve<float>(c, M * D, dry);
            L.u_at = c->prec == 0 ? (void*)L.u : carve_at(c, M * D, dry);
            if (c->prec != 0 && c->ln_fold) L.ln_part = carve<float2>(c, M * LN_PARTS, dry);
            L.z = carve_at(c, M * DM, dry);
            L.d_act = carve_at(c, M * RP, dry);
            L.h = carve_at(c, M * D, dry);
            L.soft = carve<float>(c, M, dry); L.maskf = carve<float>(c, M, dry);
            L.keep_local = carve<int>(c, M, dry); L.offsets = carve<int>(c, B, dry);
            L.total = carve<int>(c, 4, dry); L.row_src = carve<int>(c, M, dry); L.dst_of = carve<int>(c, M, dry);
        }
    }
    if (c->frames > 1) {
        c->pk_w = carve_at(c, (size_t)D * D, dry); c->pk_wT = carve_at(c, (size_t)D * D, dry);
        c->pv_w = carve_at(c, (size_t)D * D, dry); c->pv_wT = carve_at(c, (size_t)D * D, dry);
        const size_t clips = B / c->frames;
        for (int sl = 0; sl < cf.slots; ++sl) {
            PoolS& Q = c->slots[sl].pool;
            Q.xf = carve<float>(c, M * D, dry);
            Q.st_f = carve<float2>(c, M, dry); Q.st_kv = carve<float2>(c, M, dry);
            Q.xk = carve_at(c, M * D, dry); Q.xv = carve_at(c, M * D, dry);
            Q.Kp = carve_at(c, M * D, dry); Q.Vp = carve_at(c, M * D, dry);
            const size_t Mp = (M + 63) / 64 * 64;
            Q.dKt = carve_at(c, Mp * D, dry); Q.xkt = carve_at(c, Mp * D, dry);
            Q.dVt = carve_at(c, Mp * D, dry); Q.xvt = carve_at(c, Mp * D, dry);
            Q.P = carve<float>(c, B * NH * NT, dry);
            Q.o = carve<float>(c, clips * D, dry); Q.y = carve<float>(c, clips * D, dry);
            Q.dy = carve<float>(c, clips * D, dry); Q.dO = carve<float>(c, clips * D, dry);
            Q.dq_part = carve<float>(c, clips * D, dry);
            Q.qn = carve<float>(c, D, dry); Q.qhat = carve<float>(c, D, dry); Q.qs = carve<float>(c, D, dry);
            Q.gq = carve<float>(c, D, dry); Q.dqn = carve<float>(c, D, dry); Q.st_q = carve<float>(c, 4, dry);
        }
    }
    for (int sl = 0; sl < cf.slots; ++sl) {
        Transients& T = c->slots[sl].T;
        T.xn = carve_at(c, M * D, dry);
        if (c->prec != 0 && c->ln_fold) { T.st_compact = carve<float2>(c, M, dry); T.drop_src = carve<int>(c, M, dry); }
        T.h1 = carve_at(c, M * DM, dry);
        T.g_at = carve_at(c, M * D, dry);
        T.dZ = carve_at(c, M * DM, dry);
        T.ddz = carve_at(c, M * RP, dry);
        T.du_at = carve_at(c, M * D, dry);
        T.dad = c->prec != DYT_PREC_FP32 ? carve_at(c, M * D, dry) : nullptr;
        T.dact_s = c->prec != DYT_PREC_FP32 ? carve_at(c, M * RP, dry) : nullptr;
        T.dO = carve_at(c, M * D, dry);
        T.dqkv = carve_at(c, M * 3 * D, dry);
        T.dA2 = carve_at(c, M * D, dry);
        T.dxn = carve_at(c, M * D, dry);
        T.g = carve<float>(c, M * D, dry);
        T.delta = carve<float>(c, B * NH * NT, dry);
        T.dmask = carve<float>(c, M, dry);
        if (c->ad_ln) {
            T.xa = carve<float>(c, M * D, dry); T.dup = carve<float>(c, M * D, dry);
            T.aln_part = carve<float>(c, (size_t)ln_param_grad_scratch_floats((int)M), dry);
        }
        T.tok_partial = carve<float>(c, ((M + 31) / 32) * (D + 1), dry);
        T.wg_partial = carve<float>(c, ((M + 511) / 512) * (size_t)(D + 8) * 80, dry);
        T.wg_partial2 = carve<float>(c, ((M + 511) / 512) * (size_t)(D + 8) * 80, dry);
        Slot& S = c->slots[sl];
        S.dp_own = carve<float>(c, 2 * depth * B, dry);
        S.gscr = carve<float>(c, (size_t)depth * c->layer_stride, dry);
        S.wg_part.resize(depth); S.wg_part2.resize(depth); S.tok_part.resize(depth);
        for (size_t l = 0; l < depth; ++l) {
            S.wg_part[l] = carve<float>(c, ((M + 511) / 512) * (size_t)(D + 8) * 80, dry);
            S.wg_part2[l] = carve<float>(c, ((M + 511) / 512) * (size_t)(D + 8) * 80, dry);
            S.tok_part[l] = carve<float>(c, ((M + 31) / 32) * (D + 1), dry);
        }
    }
    c->grad2 = carve<float>(c, (size_t)c->n_train, dry);
    c->ad_up_bp = carve<float>(c, depth * D, dry);
    if (c->ad_ln == 2) c->ad_up_ws = carve_at(c, depth * RP * D, dry);
    c->cls_rows = carve<int>(c, B, dry);
    c->seed_dev = carve<uint64_t>(c, 2, dry);
    c->clip_scratch = carve<float>(c, 256, dry);
    c->dl_s = carve<float>(c, B * C, dry); c->dl_t = carve<float>(c, B * C, dry);
    c->logits_s = carve<float>(c, B * C, dry); c->logits_t = carve<float>(c, B * C, dry);
    c->dtok = carve<float>(c, 4, dry);
    c->loss_part = carve<float>(c, 4 * B, dry);
    c->losses = carve<float>(c, 8, dry);
}

// The aux arena (fp32 contexts, allocated by DYT_OPT_F32_SPLIT16): [hi | lo] images of the frozen matrices ([N, SPLIT_A * K] 16-bit),
// the split A-operand scratch of a pass ([M, SPLIT_A * K]) and -- bwd16 -- the 16-bit tensors a 16-bit backward pass reads.
// Carved with the main arena's helpers (the caller swaps the arena fields around the call).
static void layout_aux(dyt_ctx* c, bool dry, bool bwd16) {
    const dyt_config& cf = c->cfg;
    const size_t B = cf.max_batch, M = B * NT, depth = cf.depth, SA = SPLIT_A;
    c->arena_used = 0;
    c->pe_w3 = carve<uint16_t>(c, SA * D * D, dry);
    c->pe_w_exp = carve<int>(c, 4, dry);
    c->f8_scratch = carve<unsigned>(c, 4, dry);
    for (size_t l = 0; l < depth; ++l) {
        LayerW& w = c->W[l];
        w.w_exp = carve<int>(c, 4, dry);
        if (bwd16) {
            w.qkv_w3b = carve<uint16_t>(c, SA * 3 * D * D, dry); w.proj_w3b = carve<uint16_t>(c, SA * D * D, dry);
            w.fc1_w3b = carve<uint16_t>(c, SA * DM * D, dry); w.fc2_w3b = carve<uint16_t>(c, SA * DM * D, dry);
        }
        w.qkv_w3 = carve<uint16_t>(c, SA * 3 * D * D, dry); w.qkv_wT3 = carve<uint16_t>(c, SA * 3 * D * D, dry);
        w.proj_w3 = carve<uint16_t>(c, SA * D * D, dry); w.proj_wT3 = carve<uint16_t>(c, SA * D * D, dry);
        w.fc1_w3 = carve<uint16_t>(c, SA * DM * D, dry); w.fc1_wT3 = carve<uint16_t>(c, SA * DM * D, dry);
        w.fc2_w3 = carve<uint16_t>(c, SA * DM * D, dry); w.fc2_wT3 = carve<uint16_t>(c, SA * DM * D, dry);
        if (bwd16) {
            w.qkv_wT16 = carve<uint16_t>(c, (size_t)3 * D * D, dry); w.qkv_wTp16 = carve<uint16_t>(c, (size_t)3 * D * D, dry);
            w.proj_wT16 = carve<uint16_t>(c, (size_t)D * D, dry); w.proj_wTp16 = carve<uint16_t>(c, (size_t)D * D, dry);
            w.fc1_wT16 = carve<uint16_t>(c, (size_t)DM * D, dry); w.fc1_wTp16 = carve<uint16_t>(c, (size_t)DM * D, dry);
            w.fc2_wT16 = carve<uint16_t>(c, (size_t)DM * D, dry); w.fc2_wTp16 = carve<uint16_t>(c, (size_t)DM * D, dry);
        }
    }
    if (bwd16) {
        c->ad_up_wT16 = carve<uint16_t>(c, depth * RP * D, dry);
        c->ad_down_wT16 = carve<uint16_t>(c, depth * RP * D, dry);
        c->ad_scratch16 = carve<uint16_t>(c, 2 * depth * RP * D, dry);   // the two layouts of prep_adapters_kernel the backward does not read
        c->ad_up_w3 = carve<uint16_t>(c, SA * depth * RP * D, dry);
    }
    for (int sl = 0; sl < cf.slots; ++sl) {
        Slot& S = c->slots[sl];
        Transients& T = S.T;
        const size_t Mp = (M + 255) / 256 * 256;   // whole 256-row tiles: the fp8-correction kernel reads the rows of its last tile unclamped
        T.a3 = carve<uint16_t>(c, Mp * SA * DM, dry);
        T.xn3 = carve<uint16_t>(c, Mp * SA * D, dry);
        T.g3 = carve<uint16_t>(c, Mp * SA * D, dry);
        T.h3 = carve<uint16_t>(c, Mp * SA * DM, dry);
        T.dqkv3 = carve<uint16_t>(c, M * SA * 3 * D, dry);
        if (!bwd16) continue;
        T.dact3 = carve<uint16_t>(c, Mp * SA * RP, dry);
        T.drop_src = carve<int>(c, M, dry);   // (ln_gather writes the dropped rows' list for their three-part up-projection launch)
        T.dad16 = carve<uint16_t>(c, M * D, dry);
        T.qlo = carve<uint16_t>(c, M * D, dry); T.klo = carve<uint16_t>(c, M * D, dry); T.vlo = carve<uint16_t>(c, M * D, dry);
        S.ucls16 = carve<uint16_t>(c, B * D, dry);
        for (size_t l = 0; l < depth; ++l) {
            LayerS& L = S.L[l];
            L.q16 = carve<uint16_t>(c, M * D, dry); L.k16 = carve<uint16_t>(c, M * D, dry); L.v16 = carve<uint16_t>(c, M * D, dry);
            L.o16 = nullptr; L.ao3 = carve<uint16_t>(c, Mp * SA * D, dry);   // padded to whole 256-row tiles like the other fp8-form operand images (the 256x256 kernel reads its last tile unclamped)
             L.u16 = carve<uint16_t>(c, M * D, dry); L.h16 = carve<uint16_t>(c, M * D, dry);
            L.z16 = carve<uint16_t>(c, M * DM, dry); L.dact16 = carve<uint16_t>(c, M * RP, dry);
        }
        if (!dry) S.u0_16_own = S.L[0].u16;
    }
}
// (re)allocates the aux arena; with_bwd16: including the 16-bit backward's buffers
static int alloc_aux(dyt_ctx* c, bool with_bwd16) {
    if (c->aux_arena && (c->aux_bwd16 || !with_bwd16)) return 0;
    DYT_HIP_CHECK(hipDeviceSynchronize());
    char* main_base = c->arena; const size_t main_used = c->arena_used;
    char* old_arena = c->aux_arena; const size_t old_size = c->aux_size; const bool old_bwd16 = c->aux_bwd16;
    layout_aux(c, true, with_bwd16);   // dry run: the new size (this nulls every aux pointer; they are re-carved below on either path)
    const size_t new_size = c->arena_used;
    char* fresh = nullptr;
    // the NEW arena first: if it cannot be had, the context keeps the old one (freeing first left every W.*_w3 / T.*3 /
    // L.*16 pointer null while split16 stayed set from the earlier option value)
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&fresh), new_size);
    if (e == hipSuccess) e = hipMemset(fresh, 0, new_size);
    if (e != hipSuccess) {
        set_error("aux arena: hipMalloc / hipMemset(%zu bytes) failed: %s", new_size, hipGetErrorString(e));
        if (fresh) (void)hipFree(fresh);
        if (old_arena) { c->arena = old_arena; layout_aux(c, false, old_bwd16); }   // the old arena's layout, contents untouched
        c->arena = main_base; c->arena_used = main_used;
        c->aux_size = old_size;
        return DYT_ERR_HIP;
    }
    if (old_arena) (void)hipFree(old_arena);
    c->aux_arena = fresh;
    c->aux_size = new_size;
    c->arena = c->aux_arena;
    layout_aux(c, false, with_bwd16);
    c->arena = main_base; c->arena_used = main_used;
    c->aux_bwd16 = with_bwd16;
    DYT_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

static void trainable_layout(dyt_ctx* c) {
    const int64_t r = c->cfg.ffn_num, C = c->cfg.num_classes;
    auto a4 = [](int64_t v) { return (v + 3) / 4 * 4; };
    int64_t o = 0;
    c->off_dw = o; o = a4(o + r * D);
    c->off_db = o; o = a4(o + r);
    c->off_uw = o; o = a4(o + (int64_t)D * r);
    c->off_ub = o; o = a4(o + D);
    c->off_gw = o; o += D;       // gate weight and bias stay adjacent (one 769-wide reduction)
    c->off_gb = o; c->off_sc = o + 1; o = a4(o + 2);   // + the adapter's learnable scale (used under DYT_OPT_LEARNABLE_SCALE; a zero padding word otherwise)
    if (c->cfg.adapter_ln) { c->off_alw = o; o += D; c->off_alb = o; o += D; }   // the adapter's LayerNorm: gamma and beta adjacent (one 1536-wide reduction)
    c->layer_stride = o;
    o = c->layer_stride * c->cfg.depth;
    c->off_hw = o; o = a4(o + C * D);
    c->off_hb = o; o = a4(o + C);
    if (c->frames > 1) {
        c->off_pquery = o; o += D;
        c->off_pnq_w = o; o += D; c->off_pnq_b = o; o += D;
        c->off_pnk_w = o; o += D; c->off_pnk_b = o; o += D;   // norm_k / norm_v {w,b} stay adjacent:
        c->off_pnv_w = o; o += D; c->off_pnv_b = o; o += D;   // one 4x768 partial reduction covers them
        c->off_pq_w = o; o += (int64_t)D * D;
        c->off_pk_w = o; o += (int64_t)D * D;
        c->off_pv_w = o; o += (int64_t)D * D;
        c->off_pq_bias = o; o += D; c->off_pv_bias = o; o += D;
        c->off_pproj_w = o; o += (int64_t)D * D;
        c->off_pproj_b = o; o += D;
    }
    c->n_train = o;
}

extern "C" const char* dyt_last_error(void) { return g_err; }
extern "C" int dyt_version(void) { return 2; }   // 2: dyt_config::adapter_ln appended (round 6)
// 16-bit operand type of this build: 0 = bfloat16 (libdyt_hip.so), 1 = IEEE half (libdyt_hip_f16.so)
extern "C" int dyt_operand_type(void) {
#ifdef DYT_FP16
    return 1;
#else
    return 0;
#endif
}

extern "C" int dyt_ctx_create(const dyt_config* cfg, dyt_ctx** out) {
    if (!cfg || !out) { set_error("null argument"); return DYT_ERR_ARG; }
    if (cfg->ffn_num < 1 || cfg->ffn_num > RP || cfg->depth < 1 || cfg->depth > 64 || cfg->max_batch < 1 ||
        cfg->num_classes < 1 || cfg->num_classes > 1024 || cfg->slots < 1 || cfg->slots > 4 ||
        (cfg->precision != DYT_PREC_FP32 && cfg->precision != DYT_PREC_BF16)) {
        set_error("unsupported config: ffn_num=%d (1..64) depth=%d max_batch=%d num_classes=%d (1..1024) slots=%d precision=%d",
                  cfg->ffn_num, cfg->depth, cfg->max_batch, cfg->num_classes, cfg->slots, cfg->precision);
        return DYT_ERR_ARG;
    }
    int ndev = 0;
    DYT_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev < 1) { set_error("no HIP device"); return DYT_ERR_HIP; }
    if (cfg->frames > 1 && (cfg->max_batch % cfg->frames != 0 || cfg->frames * NT * 2 * sizeof(float) > 60 * 1024)) {
        set_error("video model: max_batch %d must be a multiple of frames %d (and frames <= 38)", cfg->max_batch, cfg->frames);
        return DYT_ERR_ARG;
    }
    if (cfg->adapter_ln < 0 || cfg->adapter_ln > 2 || (cfg->adapter_ln && cfg->frames > 1)) {
        set_error("adapter_ln=%d: 0 (none), 1 (in) or 2 (out); image model only", cfg->adapter_ln);
        return DYT_ERR_ARG;
    }
    dyt_ctx* c = new dyt_ctx();
    c->cfg = *cfg;
    c->prec = cfg->precision;
    c->frames = cfg->frames > 1 ? cfg->frames : 1;
    c->ad_ln = cfg->adapter_ln;
    if (c->frames > 1) c->cls_tail = false;   // every token of the last block reaches the pooling head
    if (c->ad_ln) c->cls_tail = false;        // (generic path: the adapter's LayerNorm runs over all rows of every block)
    c->at = at_size(c->prec);
#ifdef DYT_FP16
    if (c->prec != DYT_PREC_FP32) c->gs = 4096.0f;
#endif
    if (const char* e = getenv("DYT_LN_FOLD")) c->ln_fold = atoi(e) != 0;
    if (c->prec == 0) c->ln_fold = false;
#ifdef DYT_FP16
    c->g16 = true;   // (takes effect where the backward runs on 16-bit operands: P == 1)
    if (const char* e = getenv("DYT_G16")) c->g16 = c->g16 && atoi(e) != 0;
#endif
    trainable_layout(c);
    layout(c, true);
    c->arena_size = c->arena_used;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->arena), c->arena_size);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", c->arena_size, hipGetErrorString(e));
        delete c;
        return DYT_ERR_HIP;
    }
    layout(c, false);
    for (auto& S : c->slots) { S.u0_own = S.L[0].u; S.u0_at_own = S.L[0].u_at; S.part0_own = S.L[0].ln_part; }
    e = hipMemset(c->arena, 0, c->arena_size);
    if (e != hipSuccess) { set_error("hipMemset failed: %s", hipGetErrorString(e)); hipFree(c->arena); delete c; return DYT_ERR_HIP; }
    if (launch_cls_index(c->cls_rows, cfg->max_batch, nullptr) || hipDeviceSynchronize() != hipSuccess) {
        hipFree(c->arena); delete c; return DYT_ERR_HIP;
    }
    *out = c;
    return DYT_OK;
}

extern "C" int dyt_ctx_destroy(dyt_ctx* c) {
    if (!c) return DYT_OK;
    for (auto& r : c->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto e : c->pool) hipEventDestroy(e);
    for (auto& S : c->slots) {
        if (S.ev_f) hipEventDestroy(S.ev_f);
        if (S.ev_j) hipEventDestroy(S.ev_j);
        if (S.branch) hipStreamDestroy(S.branch);
        if (S.pool.ev_wf) hipEventDestroy(S.pool.ev_wf);
        if (S.pool.ev_wj) hipEventDestroy(S.pool.ev_wj);
        if (S.pool.wstream) hipStreamDestroy(S.pool.wstream);
    }
    if (c->ev_b0) hipEventDestroy(c->ev_b0);
    if (c->ev_half_s) hipEventDestroy(c->ev_half_s);
    if (c->ev_half_t) hipEventDestroy(c->ev_half_t);
    if (c->ev_upper) hipEventDestroy(c->ev_upper);
    if (c->ev_comm) hipEventDestroy(c->ev_comm);
    if (c->aux) hipStreamDestroy(c->aux);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->side) hipStreamDestroy(c->side);
    if (c->arena) hipFree(c->arena);
    if (c->aux_arena) hipFree(c->aux_arena);
    delete c;
    return DYT_OK;
}

extern "C" int dyt_ctx_bytes(const dyt_ctx* c, int64_t* bytes) {
    if (!c || !bytes) { set_error("null argument"); return DYT_ERR_ARG; }
    *bytes = (int64_t)(c->arena_size + c->aux_size);
    return DYT_OK;
}

extern "C" int dyt_trainable_numel(const dyt_ctx* c, int64_t* n) {
    if (!c || !n) { set_error("null argument"); return DYT_ERR_ARG; }
    *n = c->n_train;
    return DYT_OK;
}

extern "C" int dyt_trainable_offset(const dyt_ctx* c, int param, int layer, int64_t* off, int64_t* numel) {
    if (!c || !off || !numel) { set_error("null argument"); return DYT_ERR_ARG; }
    const int64_t r = c->cfg.ffn_num, C = c->cfg.num_classes;
    const int64_t base = (int64_t)layer * c->layer_stride;
    if (param >= DYT_P_AD_DOWN_W && param <= DYT_P_GATE_B && (layer < 0 || layer >= c->cfg.depth)) {
        set_error("layer %d out of range", layer);
        return DYT_ERR_ARG;
    }
    switch (param) {
        case DYT_P_AD_DOWN_W: *off = base + c->off_dw; *numel = r * D; break;
        case DYT_P_AD_DOWN_B: *off = base + c->off_db; *numel = r; break;
        case DYT_P_AD_UP_W: *off = base + c->off_uw; *numel = D * r; break;
        case DYT_P_AD_UP_B: *off = base + c->off_ub; *numel = D; break;
        case DYT_P_GATE_W: *off = base + c->off_gw; *numel = D; break;
        case DYT_P_GATE_B: *off = base + c->off_gb; *numel = 1; break;
        case DYT_P_AD_SCALE:
            if (layer < 0 || layer >= c->cfg.depth) { set_error("layer %d out of range", layer); return DYT_ERR_ARG; }
            *off = base + c->off_sc; *numel = 1; break;
        case DYT_P_AD_LN_W: case DYT_P_AD_LN_B:
            if (!c->ad_ln) { set_error("param %d exists with dyt_config.adapter_ln != 0 only", param); return DYT_ERR_ARG; }
            if (layer < 0 || layer >= c->cfg.depth) { set_error("layer %d out of range", layer); return DYT_ERR_ARG; }
            *off = base + (param == DYT_P_AD_LN_W ? c->off_alw : c->off_alb); *numel = D; break;
        case DYT_P_HEAD_W: *off = c->off_hw; *numel = C * D; break;
        case DYT_P_HEAD_B: *off = c->off_hb; *numel = C; break;
        case DYT_P_POOL_QUERY: case DYT_P_POOL_NQ_W: case DYT_P_POOL_NQ_B: case DYT_P_POOL_NK_W: case DYT_P_POOL_NK_B:
        case DYT_P_POOL_NV_W: case DYT_P_POOL_NV_B: case DYT_P_POOL_Q_W: case DYT_P_POOL_K_W: case DYT_P_POOL_V_W:
        case DYT_P_POOL_Q_BIAS: case DYT_P_POOL_V_BIAS: case DYT_P_POOL_PROJ_W: case DYT_P_POOL_PROJ_B: {
            if (c->frames <= 1) { set_error("param %d exists in the video model only (cfg.frames > 1)", param); return DYT_ERR_ARG; }
            const int64_t offs[] = {c->off_pquery, c->off_pnq_w, c->off_pnq_b, c->off_pnk_w, c->off_pnk_b, c->off_pnv_w,
                                    c->off_pnv_b, c->off_pq_w, c->off_pk_w, c->off_pv_w, c->off_pq_bias, c->off_pv_bias,
                                    c->off_pproj_w, c->off_pproj_b};
            *off = offs[param - DYT_P_POOL_QUERY];
            const bool mat = param == DYT_P_POOL_Q_W || param == DYT_P_POOL_K_W || param == DYT_P_POOL_V_W || param == DYT_P_POOL_PROJ_W;
            *numel = mat ? (int64_t)D * D : D;
            break;
        }
        default: set_error("param %d is not trainable", param); return DYT_ERR_ARG;
    }
    return DYT_OK;
}

// weight [N,K] fp32 -> AT copy and transposed AT copy [K,N]
static int set_matrix(dyt_ctx* c, const float* src, void* w, void* wT, int N, int K, hipStream_t s) {
    int rc = launch_pad_convert(c->prec, src, w, N, K, N, K, s);
    if (rc) return rc;
    if (wT) rc = launch_transpose_convert(c->prec, src, wT, N, K, K, N, s);
    return rc;
}
// fp32 mode: refresh the 16-bit [hi | lo] images of one layer's (layer < 0: the patch embedding's) frozen matrices
static int refresh_split(dyt_ctx* c, int layer, hipStream_t s) {
    if (c->prec != 0) return 0;
    if (c->f8) {   // forward images per class in the hi16 / fp8 or the [hi | lo] form (the backward of these modes runs on the 16-bit copies: no transposed images)
        const int fm = c->f8_mask, fmc = c->f8_mask_complete;
        auto one = [&](int bit, const void* src, void* dst, int N, int K, int* ew, void* dst_b = nullptr) {
            int rc = (fm & bit) ? launch_split_w_f8((const float*)src, dst, N, K, ew, c->f8_scratch, s) : launch_split3_w((const float*)src, dst, N, K, s);
            if (!rc && dst_b && ((fm ^ fmc) & bit))   // the complete_model pass takes the other form of this class: its own image
                rc = (fmc & bit) ? launch_split_w_f8((const float*)src, dst_b, N, K, ew, c->f8_scratch, s) : launch_split3_w((const float*)src, dst_b, N, K, s);
            return rc;
        };
        if (layer < 0) return one(16, c->pe_w, c->pe_w3, D, D, c->pe_w_exp);
        LayerW& w = c->W[layer];
        int rc = one(1, w.qkv_w, w.qkv_w3, 3 * D, D, w.w_exp + 0, w.qkv_w3b);
        if (!rc) rc = one(2, w.proj_w, w.proj_w3, D, D, w.w_exp + 1, w.proj_w3b);
        if (!rc) rc = one(4, w.fc1_w, w.fc1_w3, DM, D, w.w_exp + 2, w.fc1_w3b);
        if (!rc) rc = one(8, w.fc2_w, w.fc2_w3, D, DM, w.w_exp + 3, w.fc2_w3b);
        return rc;
    }
    if (layer < 0) return launch_split3_w((const float*)c->pe_w, c->pe_w3, D, D, s);
    LayerW& w = c->W[layer];
    int rc = launch_split3_w((const float*)w.qkv_w, w.qkv_w3, 3 * D, D, s);
    if (!rc) rc = launch_split3_w((const float*)w.qkv_wT, w.qkv_wT3, D, 3 * D, s);
    if (!rc) rc = launch_split3_w((const float*)w.proj_w, w.proj_w3, D, D, s);
    if (!rc) rc = launch_split3_w((const float*)w.proj_wT, w.proj_wT3, D, D, s);
    if (!rc) rc = launch_split3_w((const float*)w.fc1_w, w.fc1_w3, DM, D, s);
    if (!rc) rc = launch_split3_w((const float*)w.fc1_wT, w.fc1_wT3, D, DM, s);
    if (!rc) rc = launch_split3_w((const float*)w.fc2_w, w.fc2_w3, D, DM, s);
    if (!rc) rc = launch_split3_w((const float*)w.fc2_wT, w.fc2_wT3, DM, D, s);
    return rc;
}
// bwd16: the 16-bit transposed copies (plain + MFMA fragment order) of one layer's matrices, from the fp32 transposed copies
static int refresh_bwd16(dyt_ctx* c, int layer, hipStream_t s) {
    if (c->prec != 0 || !c->bwd16 || layer < 0) return 0;
    LayerW& w = c->W[layer];
    struct M16 { const void* src; void* dst; void* dstp; int N, K; };
    const M16 m[4] = {{w.qkv_wT, w.qkv_wT16, w.qkv_wTp16, D, 3 * D}, {w.proj_wT, w.proj_wT16, w.proj_wTp16, D, D},
                      {w.fc1_wT, w.fc1_wT16, w.fc1_wTp16, D, DM}, {w.fc2_wT, w.fc2_wT16, w.fc2_wTp16, DM, D}};
    for (const M16& x : m) {
        int rc = launch_convert(1, (const float*)x.src, x.dst, (int64_t)x.N * x.K, s);
        if (!rc) rc = launch_preshuffle_w(x.dst, x.dstp, x.N, x.K, s);
        if (rc) return rc;
    }
    return 0;
}
// 16-bit modes: the gamma-folded fc1 weight, its column sums and the beta-folded bias of one layer, from what the layer holds NOW (called
// after every upload of one of the four tensors involved: the last one leaves them consistent)
static int refresh_ln_fold(dyt_ctx* c, int layer, hipStream_t s) {
    LayerW& w = c->W[layer];
    if (c->prec == 0 || !w.fc1_wf) return 0;
    return launch_ln_fold_w(c->prec, w.fc1_w32, w.ln2_w, w.ln2_b, w.fc1_b, w.fc1_wf, w.fc1_wfp, w.fc1_cs, w.fc1_bf, DM, s);
}
static int copy_f32(float* dst, const float* src, size_t n, hipStream_t s) {
    DYT_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

static int set_frozen_impl(dyt_ctx* c, int param, int layer, const float* src, void* stream);
extern "C" int dyt_set_frozen(dyt_ctx* c, int param, int layer, const float* src, void* stream) {
    int rc = set_frozen_impl(c, param, layer, src, stream);
    if (!rc && c->split16 && (param == DYT_P_PE_W || param == DYT_P_QKV_W || param == DYT_P_PROJ_W || param == DYT_P_FC1_W || param == DYT_P_FC2_W)) {
        rc = refresh_split(c, param == DYT_P_PE_W ? -1 : layer, static_cast<hipStream_t>(stream));
        if (!rc && param != DYT_P_PE_W) rc = refresh_bwd16(c, layer, static_cast<hipStream_t>(stream));
    }
    return rc;
}
static int set_frozen_impl(dyt_ctx* c, int param, int layer, const float* src, void* stream) {
    if (!c || !src) { set_error("null argument"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool per_layer = param >= DYT_P_LN1_W && param <= DYT_P_FC2_B;
    if (per_layer && (layer < 0 || layer >= c->cfg.depth)) { set_error("layer %d out of range", layer); return DYT_ERR_ARG; }
    LayerW* w = per_layer ? &c->W[layer] : nullptr;
    switch (param) {
        case DYT_P_CLS: return copy_f32(c->cls, src, D, s);
        case DYT_P_POS: return copy_f32(c->pos, src, (size_t)NT * D, s);
        case DYT_P_PE_W: return set_matrix(c, src, c->pe_w, nullptr, D, D, s);  // [768, 3*16*16]
        case DYT_P_PE_B: return copy_f32(c->pe_b, src, D, s);
        case DYT_P_LN1_W: return copy_f32(w->ln1_w, src, D, s);
        case DYT_P_LN1_B: return copy_f32(w->ln1_b, src, D, s);
        case DYT_P_QKV_W: {
            int rc = set_matrix(c, src, w->qkv_w, w->qkv_wT, 3 * D, D, s);
            if (!rc && w->qkv_wp) rc = launch_preshuffle_w(w->qkv_w, w->qkv_wp, 3 * D, D, s);
            if (!rc && w->qkv_wTp) rc = launch_preshuffle_w(w->qkv_wT, w->qkv_wTp, D, 3 * D, s);
            return rc;
        }
        case DYT_P_QKV_B: return copy_f32(w->qkv_b, src, 3 * D, s);
        case DYT_P_PROJ_W: {
            int rc = set_matrix(c, src, w->proj_w, w->proj_wT, D, D, s);
            if (!rc && w->proj_wp) rc = launch_preshuffle_w(w->proj_w, w->proj_wp, D, D, s);
            if (!rc && w->proj_wTp) rc = launch_preshuffle_w(w->proj_wT, w->proj_wTp, D, D, s);
            return rc;
        }
        case DYT_P_PROJ_B: return copy_f32(w->proj_b, src, D, s);
        case DYT_P_LN2_W: { int rc = copy_f32(w->ln2_w, src, D, s); return rc ? rc : refresh_ln_fold(c, layer, s); }
        case DYT_P_LN2_B: { int rc = copy_f32(w->ln2_b, src, D, s); return rc ? rc : refresh_ln_fold(c, layer, s); }
        case DYT_P_FC1_W: {
            int rc = set_matrix(c, src, w->fc1_w, w->fc1_wT, DM, D, s);
            if (!rc && w->fc1_wp) rc = launch_preshuffle_w(w->fc1_w, w->fc1_wp, DM, D, s);
            if (!rc && w->fc1_wTp) rc = launch_preshuffle_w(w->fc1_wT, w->fc1_wTp, D, DM, s);
            if (!rc && w->fc1_w32) rc = copy_f32(w->fc1_w32, src, (size_t)DM * D, s);
            return rc ? rc : refresh_ln_fold(c, layer, s);
        }
        case DYT_P_FC1_B: { int rc = copy_f32(w->fc1_b, src, DM, s); return rc ? rc : refresh_ln_fold(c, layer, s); }
        case DYT_P_FC2_W: {
            int rc = set_matrix(c, src, w->fc2_w, w->fc2_wT, D, DM, s);
            if (!rc && w->fc2_wTp) rc = launch_preshuffle_w(w->fc2_wT, w->fc2_wTp, DM, D, s);   // fc2^T: [3072, 768]
            return rc;
        }
        case DYT_P_FC2_B: return copy_f32(w->fc2_b, src, D, s);
        case DYT_P_NORM_W: return copy_f32(c->norm_w, src, D, s);
        case DYT_P_NORM_B: return copy_f32(c->norm_b, src, D, s);
        default: set_error("param %d is not a frozen parameter", param); return DYT_ERR_ARG;
    }
}

// ------------------------------------------------------------------------------------------
// profiling helpers
// ------------------------------------------------------------------------------------------
static hipEvent_t get_event(dyt_ctx* c) {
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
struct ProfScope {
    dyt_ctx* c; hipStream_t s; ProfRec r; bool on;
    ProfScope(dyt_ctx* c_, hipStream_t s_, int cat, double flops, const int* m_dev = nullptr, int M = 0)
        : c(c_), s(s_), on(c_->prof) {
        if (on) {
            r.cat = cat; r.flops = flops; r.m_dev = m_dev; r.M = M;
            r.a = get_event(c); r.b = get_event(c);
            hipEventRecord(r.a, s);
        }
    }
    ~ProfScope() { if (on) { hipEventRecord(r.b, s); c->recs.push_back(r); } }
};
#define RUN(cat, flops, call)                  \
    do {                                       \
        ProfScope _ps(c, s, (cat), (flops));   \
        int _rc = (call);                      \
        if (_rc) return _rc;                   \
    } while (0)
// GEMM whose valid row count lives on the device (compacted MLP): FLOPs are scaled by the
// count actually processed when the profile is read back
#define RUN_GEMM(kind, a)                                                \
    do {                                                                 \
        ProfScope _ps(c, s, 0, (a).flops(), (a).m_dev, (a).M);           \
        int _rc = launch_gemm(P, (kind), (a), s);                        \
        if (_rc) return _rc;                                             \
    } while (0)

extern "C" int dyt_ctx_set_option(dyt_ctx* c, int option, int value) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    switch (option) {
        case DYT_OPT_STREAM_OVERLAP:   // 0 off, 1 on (the two passes), 2 passes + adapter branches, 3 adapter branches only, 4 forward passes only
            c->overlap = value != 0; c->ov_pass = value == 1 || value == 2 || value == 4; c->ov_branch = value == 2 || value == 3;
            c->ov_bwd_serial = value == 4;
            return DYT_OK;
        case DYT_OPT_CLS_TAIL: c->cls_tail = value != 0 && c->frames <= 1 && !c->ad_ln; for (auto& S : c->slots) S.valid = false; return DYT_OK;
        case DYT_OPT_SHARE_BLOCK0: c->share_block0 = value != 0; return DYT_OK;
        case DYT_OPT_GRAD_SCALE_LOG2:   // 16-bit gradient operands carry 2^value (0 = none); fp32 mode ignores it
            if (value < 0 || value > 24) { set_error("grad scale log2 %d out of range 0..24", value); return DYT_ERR_ARG; }
            c->gs = (c->prec == DYT_PREC_FP32 && !c->bwd16) ? 1.0f : (float)(1u << value);
            return DYT_OK;
        case DYT_OPT_FC2_CAT: c->fc2_cat = value != 0; return DYT_OK;
        case DYT_OPT_F32_SPLIT16: {   // fp32 mode only: the frozen-weight GEMMs as hi*hi + hi*lo + lo*hi on the 16-bit matrix cores
            if (c->prec != 0) { set_error("DYT_OPT_F32_SPLIT16 applies to the fp32 mode"); return DYT_ERR_ARG; }
            if (value < 0 || value > 5) { set_error("DYT_OPT_F32_SPLIT16: value %d (0 off, 1 every product three-part, 2 gradient products one-part, 3 backward on 16-bit operands, 4 = 3 with fp8 correction products)", value); return DYT_ERR_ARG; }
            if (value != 0) { int rc = alloc_aux(c, value >= 3); if (rc) return rc; }
            c->split16 = value != 0;
            // 3 ("fp16x3h"): the forward as in 1 / 2 bit for bit, with what the backward needs saved in the 16-bit operand type (the hi
            // parts the forward computes anyway; ReLU / dropout / gate masks are the exact forward's), and the backward pass on the
            // 16-bit mode's data flow and kernels (fused attention backward, pre-shuffled-weight dgrads, 16-bit weight gradients)
            c->bwd16 = value >= 3;
            // 4 ("fp16f8"): the forward GEMMs' two correction products hi * lo + lo * hi on the fp8 matrix cores (twice the f16 rate per k:
            // 2K- instead of 3K-equivalent contractions; per-GEMM error ~2^-15 instead of 2^-20, logits ~5e-5 from the fp32 reference)
            c->f8 = value >= 4;
            // 5 ("fp16x3q"): only the attention branch's GEMMs (qkv, proj) that way -- the MLP's K = 3072 contractions and the GELU
            // between them carry most of the fp8 form's error: gate logits stay at the three-part level (emulated 1.0e-5 vs 4.4e-6 / 4.7e-5)
            c->f8_mask = value == 4 ? 31 : (value == 5 ? 3 : 0);
            // ... and the complete_model (teacher) pass of 5 takes the fp8-correction form for the MLP as well: its gate output is discarded, no
            // token-keep decision depends on it, only its logits (5e-5 from the reference instead of 7e-6) -- the dense pass is the heavier one
            c->f8_mask_complete = value == 4 ? 31 : (value == 5 ? 15 + 32 : 0);   // (32: that pass's attention forward as the hi * hi product alone)
            if (const char* e = getenv("DYT_F8_CLASSES")) { if (c->f8) c->f8_mask = c->f8_mask_complete = atoi(e) & 31; }   // measurement knobs
            if (const char* e = getenv("DYT_F8_CLASSES_COMPLETE")) { if (c->f8) c->f8_mask_complete = atoi(e) & 63; }   // (32: that pass's attention forward as the hi * hi product alone)
            c->f8_mask_complete = (c->f8_mask_complete & ~16) | (c->f8_mask & 16);   // the patch embedding is shared between the passes
            c->gs = 1.0f;
#ifdef DYT_FP16
            if (c->bwd16) c->gs = 4096.0f;   // the fixed loss scale of the fp16 mode (dyt_ctx::gs)
#endif
            // 2 ("fp16x3f"): the forward (logits, gate decisions, losses, saved activations) as in 1; the gradient GEMMs contract
            // dY_hi * W_hi alone and the attention backward's dP / dQ / dK / dV take the hi * hi product (its score recomputation keeps three)
            c->split_bwd_parts = c->split_bwd_attn_parts = value == 2 ? 1 : 3;
            if (const char* e = getenv("DYT_SPLIT_GS_LOG2")) c->split_gs = (float)(1u << atoi(e));   // measurement knob
            if (const char* e = getenv("DYT_SPLIT_BWD_PARTS")) c->split_bwd_parts = std::min(3, std::max(1, atoi(e)));
            if (const char* e = getenv("DYT_SPLIT_BWD_ATTN_PARTS")) c->split_bwd_attn_parts = atoi(e) >= 3 ? 3 : 1;
            if (const char* e = getenv("DYT_SPLIT_FWD_PARTS")) sscanf(e, "%d,%d,%d,%d", &c->split_fwd_parts[0], &c->split_fwd_parts[1], &c->split_fwd_parts[2], &c->split_fwd_parts[3]);
            if (const char* e = getenv("DYT_SPLIT_WGRAD16")) c->split_wgrad16 = atoi(e) != 0;
            if (const char* e = getenv("DYT_SPLIT_ATTN")) c->split_attn = atoi(e) != 0;
            if (const char* e = getenv("DYT_SPLIT_PROD")) c->split_prod = atoi(e) != 0;
            if (c->bwd16) c->split_attn = c->split_prod = true;   // the 16-bit backward reads what the split attention kernel / producers write (planes, the per-layer proj operand image)
            for (auto& S : c->slots) S.valid = false;
            if (c->split16) {   // parts of the weights uploaded so far (later dyt_set_frozen calls refresh theirs)
                DYT_HIP_CHECK(hipDeviceSynchronize());   // uploads may be in flight on the caller's streams
                int rc = refresh_split(c, -1, nullptr);
                for (int l = 0; l < c->cfg.depth && !rc; ++l) { rc = refresh_split(c, l, nullptr); if (!rc) rc = refresh_bwd16(c, l, nullptr); }
                if (rc) return rc;
                DYT_HIP_CHECK(hipDeviceSynchronize());
            }
            return DYT_OK;
        }
        case DYT_OPT_ATTN_BWD_FUSED: set_attn_bwd_fused(value); return DYT_OK;   // process-wide
        case DYT_OPT_ATTN_V2: set_attn_v2(value & 3); return DYT_OK;             // process-wide
        case DYT_OPT_GEMM_SPLITK: set_gemm_splitk(value); return DYT_OK;         // process-wide
        case DYT_OPT_LEARNABLE_SCALE:
            // The scale words are the CALLER's (DYT_P_AD_SCALE of the flat trainable buffer handed to every call): a padding word -- zero -- until
            // the caller writes the reference's initial value 1.0 (models/dynamic_adapter.py:102).  Switching the meaning of that word between
            // passes would silently turn the adapters off (s = 0) or rescale them, so the option is fixed once a pass has run.
            if (value != 0 && c->ad_ln) { set_error("learnable adapter scale together with the adapter's LayerNorm option is not supported"); return DYT_ERR_ARG; }
            if ((value != 0) != c->learn_scale && c->pass_ran) {
                set_error("DYT_OPT_LEARNABLE_SCALE must be set before the first forward pass of the context");
                return DYT_ERR_STATE;
            }
            c->learn_scale = value != 0; for (auto& S : c->slots) S.valid = false; return DYT_OK;
        case DYT_OPT_COUNT_FLOPS_TOKENS:
            if (value < 0 || value > NT) { set_error("count_flops tokens %d out of range 0..197", value); return DYT_ERR_ARG; }
            c->count_flops_tokens = value; for (auto& S : c->slots) S.valid = false; return DYT_OK;
    }
    set_error("unknown option %d", option);
    return DYT_ERR_ARG;
}

extern "C" int dyt_set_drop_path(dyt_ctx* c, float rate) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    if (!(rate >= 0.f) || rate >= 1.f) { set_error("drop_path rate %g out of [0, 1)", rate); return DYT_ERR_ARG; }
    c->drop_path_rate = rate;
    return DYT_OK;
}
extern "C" int dyt_set_drop_path_scales(dyt_ctx* c, int slot, const float* scales) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    if (slot < 0 || slot >= c->cfg.slots) { set_error("slot %d out of range", slot); return DYT_ERR_ARG; }
    c->slots[slot].dp_inject = scales;
    return DYT_OK;
}

extern "C" int dyt_set_soft_targets(dyt_ctx* c, const float* targets, int rows) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    if (targets && rows < 1) { set_error("soft targets: rows = %d", rows); return DYT_ERR_ARG; }
    c->soft_targets = targets; c->soft_batch = targets ? rows : 0;
    return DYT_OK;
}

extern "C" int dyt_set_global_option(int option, int value) {
    if (option == DYT_OPT_ATTN_BWD_FUSED) { set_attn_bwd_fused(value); return DYT_OK; }
    if (option == DYT_OPT_ATTN_V2) { set_attn_v2(value & 3); return DYT_OK; }
    if (option == DYT_OPT_GEMM_SPLITK) { set_gemm_splitk(value); return DYT_OK; }
    if (option == DYT_OPT_F32_SPLIT16) { set_attn_f32_split(value); return DYT_OK; }   // unit entry dyt_attention(precision 0): split forward kernel
    set_error("option %d is not process-wide", option);
    return DYT_ERR_ARG;
}

extern "C" int dyt_profile_enable(dyt_ctx* c, int on) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    c->prof = on != 0;
    if (on) gemm_kernel_launch_count(1);
    return DYT_OK;
}
extern "C" int dyt_profile_read(dyt_ctx* c, int category, double* ms, int64_t* launches, double* flops) {
    if (!c || !ms || !launches || !flops) { set_error("null argument"); return DYT_ERR_ARG; }
    if (category == 3) {   // bf16 GEMM KERNEL launches since dyt_profile_enable(ctx, 1) (a GEMM may take two)
        *ms = 0; *flops = 0; *launches = gemm_kernel_launch_count(0);
        return DYT_OK;
    }
    DYT_HIP_CHECK(hipDeviceSynchronize());
    double t = 0, f = 0;
    int64_t n = 0;
    std::vector<ProfRec> keep;
    for (auto& r : c->recs) {
        if (r.cat != category) { keep.push_back(r); continue; }
        float e = 0.f;
        hipEventElapsedTime(&e, r.a, r.b);
        double fl = r.flops;
        if (r.m_dev && r.M > 0) {  // rows really processed (still holds the last step's count)
            int m = r.M;
            if (hipMemcpy(&m, r.m_dev, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && m < r.M) fl *= (double)m / r.M;
        }
        t += e; f += fl; ++n;
        c->pool.push_back(r.a); c->pool.push_back(r.b);
    }
    c->recs.swap(keep);
    *ms = t; *launches = n; *flops = f;
    return DYT_OK;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
static int prep_adapters(dyt_ctx* c, const float* trainable, hipStream_t s) {
    const dim3 grid((RP * D + 255) / 256, c->cfg.depth);
    const int64_t sc_off = c->learn_scale ? c->off_sc : -1;
    // "out" form of the adapter's LayerNorm: its input is s (d_act W_up^T + b_up) -- scaled weight copies + a scaled bias copy for that GEMM
    const bool out_ln = c->ad_ln == 2;
    float* up_bp = (c->learn_scale || out_ln) ? c->ad_up_bp : nullptr;
    const float ws_scale = out_ln ? c->cfg.adapter_scale : 0.f, b_scale = out_ln ? c->cfg.adapter_scale : 1.0f;
    if (c->prec == 0) {
        hipLaunchKernelGGL(prep_adapters_kernel<float>, grid, dim3(256), 0, s, trainable, c->layer_stride, c->off_dw, c->off_db,
                           c->off_uw, c->cfg.ffn_num, (float*)c->ad_down_w, (float*)c->ad_down_wT, (float*)c->ad_up_w,
                           (float*)c->ad_up_wT, c->ad_down_b, out_ln ? (float*)c->ad_up_ws : (float*)nullptr, ws_scale, sc_off, c->off_ub, up_bp, b_scale);
        if (c->bwd16) {   // + the 16-bit transposes the 16-bit backward's adapter dgrads multiply by
            bf16* scr = (bf16*)c->ad_scratch16;
            hipLaunchKernelGGL(prep_adapters_kernel<bf16>, grid, dim3(256), 0, s, trainable, c->layer_stride, c->off_dw, c->off_db,
                               c->off_uw, c->cfg.ffn_num, scr, (bf16*)c->ad_down_wT16, scr + (size_t)c->cfg.depth * RP * D,
                               (bf16*)c->ad_up_wT16, c->ad_down_b, (bf16*)nullptr, 0.f, sc_off, c->off_ub, up_bp, b_scale);
            if (c->ad_up_w3) {   // [hi | lo] image of the fp32 up-projection copies just written (all blocks: depth * 768 rows of 64)
                int rc = launch_split3_w((const float*)c->ad_up_w, c->ad_up_w3, c->cfg.depth * D, RP, s);
                if (rc) return rc;
            }
        }
    } else
        hipLaunchKernelGGL(prep_adapters_kernel<bf16>, grid, dim3(256), 0, s, trainable, c->layer_stride, c->off_dw, c->off_db,
                           c->off_uw, c->cfg.ffn_num, (bf16*)c->ad_down_w, (bf16*)c->ad_down_wT, (bf16*)c->ad_up_w,
                           (bf16*)c->ad_up_wT, c->ad_down_b, out_ln ? (bf16*)c->ad_up_ws : (bf16*)nullptr, ws_scale, sc_off, c->off_ub, up_bp, b_scale);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

// adapter-branch side stream of a pass (created on first use); null when overlap is off / profiling
static int branch_stream(dyt_ctx* c, Slot& S, hipStream_t* out) {
    *out = nullptr;
    if (!c->overlap || !c->ov_branch || c->prof) return 0;
    if (S.no_branch) return 0;
    if (!S.branch) {
        DYT_HIP_CHECK(hipStreamCreateWithFlags(&S.branch, hipStreamNonBlocking));
        DYT_HIP_CHECK(hipEventCreateWithFlags(&S.ev_f, hipEventDisableTiming));
        DYT_HIP_CHECK(hipEventCreateWithFlags(&S.ev_j, hipEventDisableTiming));
    }
    *out = S.branch;
    return 0;
}
#define FORK(sb)  do { if (sb) { DYT_HIP_CHECK(hipEventRecord(S.ev_f, s)); DYT_HIP_CHECK(hipStreamWaitEvent(sb, S.ev_f, 0)); } } while (0)
#define JOIN(sb)  do { if (sb) { DYT_HIP_CHECK(hipEventRecord(S.ev_j, sb)); DYT_HIP_CHECK(hipStreamWaitEvent(s, S.ev_j, 0)); } } while (0)
// run a launch on the branch stream when there is one
#define RUN_ON(sb, cat, flops, call)           \
    do {                                       \
        hipStream_t _keep = s;                 \
        if (sb) s = sb;                        \
        ProfScope _ps(c, s, (cat), (flops));   \
        int _rc = (call);                      \
        s = _keep;                             \
        if (_rc) return _rc;                   \
    } while (0)

static inline void* at_off(const dyt_ctx* c, void* base, size_t elems) { return static_cast<char*>(base) + elems * c->at; }
#define SPLIT(a, w3) do { if (c->split16) { (a).W3 = (w3); (a).a3 = T.a3; } } while (0)
// forward GEMM class g (0 qkv, 1 proj, 2 fc1, 3 fc2): products of its contraction (measurement knob DYT_SPLIT_FWD_PARTS="qkv,proj,fc1,fc2")
// (w3b: the class's second image, used by the pass whose form differs from the student's)
// one_part (bit g; complete_model passes only): the class's GEMM as the hi * hi product alone -- both operand images carry the IEEE-half hi part in
// front of either form's second part (same row stride), so the image the pass would read anyway serves
#define SPLIT_F(a, w3, w3b, g) do { SPLIT(a, (((fm ^ c->f8_mask) >> (g)) & 1) ? (w3b) : (w3)); if (c->split16) (a).a3_parts = c->split_fwd_parts[g]; \
        if (c->split16 && ((one_part >> (g)) & 1)) (a).a3_parts = 1; \
        else if ((fm >> (g)) & 1) { (a).f8 = true; (a).w_exp = W.w_exp + (g); } } while (0)
// gradient operands: scaled by 2^12 before the split so that the lo parts stay fp16 normals (the loss scale of the fp16 mode)
// the producing kernel already wrote the split operand into `buf`
#define SPLIT_READY(a, buf) do { if (c->split16) { (a).a3 = (buf); (a).a3_ready = true; } } while (0)
#define SPLIT_G(a, w3) do { if (c->split16) { (a).W3 = (w3); (a).a3 = T.a3; (a).a3_scale = c->split_gs; (a).a3_parts = c->split_bwd_parts; } } while (0)

// ------------------------------------------------------------------------------------------
// video model: attentive pooling head (video_models/video_vision_transformer_IN21K.py:463-483)
// ------------------------------------------------------------------------------------------
static int prep_pool(dyt_ctx* c, const float* tr, hipStream_t s) {
    if (c->frames <= 1) return 0;
    int rc = set_matrix(c, tr + c->off_pk_w, c->pk_w, c->pk_wT, D, D, s);
    if (rc) return rc;
    return set_matrix(c, tr + c->off_pv_w, c->pv_w, c->pv_wT, D, D, s);
}

static int pool_forward(dyt_ctx* c, Slot& S, const float* tr, float* logits, int B, hipStream_t s) {
    const int P = c->prec, t = c->frames, clips = B / t, M = B * NT, NK = t * NT, C = c->cfg.num_classes;
    PoolS& Q = S.pool;
    RUN(2, 0, launch_pool_ln_fwd(P, S.xs[c->cfg.depth], c->norm_w, c->norm_b, tr + c->off_pnk_w, tr + c->off_pnk_b,
                                 tr + c->off_pnv_w, tr + c->off_pnv_b, Q.xf, Q.st_f, Q.st_kv, Q.xk, Q.xv, M, s));
    {
        GemmArgs a; a.A = Q.xk; a.W = c->pk_w; a.M = M; a.N = D; a.K = D; a.out_at = Q.Kp;   // k has no bias
        RUN_GEMM(EPI_BIAS_AT, a);
    }
    {
        GemmArgs a; a.A = Q.xv; a.W = c->pv_w; a.M = M; a.N = D; a.K = D; a.bias = tr + c->off_pv_bias; a.out_at = Q.Vp;
        RUN_GEMM(EPI_BIAS_AT, a);
    }
    RUN(2, 0, launch_pool_q_fwd(tr + c->off_pquery, tr + c->off_pnq_w, tr + c->off_pnq_b, tr + c->off_pq_w,
                                tr + c->off_pq_bias, Q.qn, Q.qhat, Q.st_q, Q.qs, s));
    RUN(1, 4.0 * clips * NH * (double)NK * HD, launch_pool_attn_fwd(P, Q.qs, Q.Kp, Q.Vp, Q.P, Q.o, clips, NK, s));
    RUN(2, 0, launch_rows_linear(Q.o, tr + c->off_pproj_w, tr + c->off_pproj_b, Q.y, clips, D, D, 1.0f, s));
    RUN(2, 0, launch_rows_linear(Q.y, tr + c->off_hw, tr + c->off_hb, logits, clips, C, D, 1.0f, s));
    return 0;
}

// dlogits [clips, C] -> pooling-head weight gradients (accumulated into grad) and T.g = dL/d x_last [M,768]
static int pool_backward(dyt_ctx* c, Slot& S, const float* tr, const float* dlogits, float* grad, hipStream_t s) {
    const int P = c->prec, t = c->frames, B = S.batch, clips = B / t, M = B * NT, NK = t * NT, C = c->cfg.num_classes;
    const int Mp = (M + 63) / 64 * 64;
    PoolS& Q = S.pool;
    Transients& T = S.T;
    const float gs = P == 0 ? 1.0f : c->gs, inv_gs = 1.0f / gs;   // 16-bit gradient operands carry gs (fp16 build)
    RUN(2, 0, launch_rows_linear_bwd(dlogits, Q.y, tr + c->off_hw, Q.dy, grad + c->off_hw, grad + c->off_hb, clips, C, D, s));
    RUN(2, 0, launch_rows_linear_bwd(Q.dy, Q.o, tr + c->off_pproj_w, Q.dO, grad + c->off_pproj_w, grad + c->off_pproj_b,
                                     clips, D, D, s));
    void* dK = T.dO; void* dV = T.dxn;   // [M,768] AT transients, free until the trunk backward starts
    RUN(1, 8.0 * clips * NH * (double)NK * HD,
        launch_pool_attn_bwd(P, Q.qs, Q.Kp, Q.Vp, Q.P, Q.dO, dK, dV, Q.dq_part, clips, NK, gs, s));
    RUN(2, 0, launch_pool_q_bwd(Q.dq_part, clips, Q.qn, Q.qhat, Q.st_q, tr + c->off_pq_w, tr + c->off_pnq_w, Q.gq, Q.dqn,
                                grad + c->off_pq_w, grad + c->off_pq_bias, grad + c->off_pnq_w, grad + c->off_pnq_b,
                                grad + c->off_pquery, s));
    // weight gradients of k / v: dW = dK^T xk over the token rows -- operands transposed to K-contiguous, NT GEMM.
    // A 768x768 output is only 36 workgroups, so the two GEMMs go to a side stream and run under the trunk's
    // backward (they touch nothing else: their operands are private copies, their outputs own regions of grad).
    RUN(2, 0, launch_transpose_rows(P, dK, Q.dKt, M, Mp, nullptr, s));
    RUN(2, 0, launch_transpose_rows(P, Q.xk, Q.xkt, M, Mp, nullptr, s));
    RUN(2, 0, launch_transpose_rows(P, dV, Q.dVt, M, Mp, T.tok_partial, s));   // + column sums of dV -> v_bias
    RUN(2, 0, launch_transpose_rows(P, Q.xv, Q.xvt, M, Mp, nullptr, s));
    RUN(2, 0, launch_reduce_partials(T.tok_partial, Mp / 64, D, grad + c->off_pv_bias, D, inv_gs, s));
    hipStream_t ws = nullptr;
    if (c->overlap && !c->prof) {
        if (!Q.wstream) {
            DYT_HIP_CHECK(hipStreamCreateWithFlags(&Q.wstream, hipStreamNonBlocking));
            DYT_HIP_CHECK(hipEventCreateWithFlags(&Q.ev_wf, hipEventDisableTiming));
            DYT_HIP_CHECK(hipEventCreateWithFlags(&Q.ev_wj, hipEventDisableTiming));
        }
        ws = Q.wstream;
        DYT_HIP_CHECK(hipEventRecord(Q.ev_wf, s));
        DYT_HIP_CHECK(hipStreamWaitEvent(ws, Q.ev_wf, 0));
    }
    {
        GemmArgs a; a.A = Q.dKt; a.W = Q.xkt; a.M = D; a.N = D; a.K = Mp; a.out_f32 = grad + c->off_pk_w; a.accumulate = 1; a.scale = inv_gs;
        RUN_ON(ws, 0, a.flops(), launch_gemm(P, EPI_STORE_F32, a, s));
    }
    {
        GemmArgs a; a.A = Q.dVt; a.W = Q.xvt; a.M = D; a.N = D; a.K = Mp; a.out_f32 = grad + c->off_pv_w; a.accumulate = 1; a.scale = inv_gs;
        RUN_ON(ws, 0, a.flops(), launch_gemm(P, EPI_STORE_F32, a, s));
    }
    if (ws) { DYT_HIP_CHECK(hipEventRecord(Q.ev_wj, ws)); Q.wpending = true; }   // joined at the end of backward_impl
    // dgrads through k / v, then norm_k + norm_v + final norm backward in one row pass
    {
        GemmArgs a; a.A = dK; a.W = c->pk_wT; a.M = M; a.N = D; a.K = D; a.out_at = T.du_at;
        RUN_GEMM(EPI_STORE_AT, a);
    }
    {
        GemmArgs a; a.A = dV; a.W = c->pv_wT; a.M = M; a.N = D; a.K = D; a.out_at = T.dA2;
        RUN_GEMM(EPI_STORE_AT, a);
    }
    int nblk = 0;
    RUN(2, 0, launch_pool_ln_bwd(P, T.du_at, T.dA2, Q.xf, Q.st_kv, tr + c->off_pnk_w, tr + c->off_pnv_w, S.xs[c->cfg.depth],
                                 Q.st_f, c->norm_w, T.g, T.wg_partial, M, &nblk, gs, s));
    RUN(2, 0, launch_reduce_partials(T.wg_partial, nblk, 4 * D, grad + c->off_pnk_w, 4 * D, 1.0f, s));
    return 0;
}

static int forward_impl(dyt_ctx* c, int slot, const float* images, int B, int flags, const float* trainable,
                        const float* g1, const float* g2, const uint8_t* keep_mask, uint64_t seed, float* logits,
                        float* token_select, float* token_logits, bool do_prep, hipStream_t s,
                        const Slot* share0 = nullptr, hipEvent_t ev_b0_record = nullptr, hipEvent_t ev_b0_wait = nullptr) {
    // share0: reuse another pass's embedding + block-0 attention branch (same images, same frozen weights):
    //         its u / u_at of block 0 become this pass's (the step function aliases the pointers).
    if (slot < 0 || slot >= c->cfg.slots) { set_error("slot %d out of range", slot); return DYT_ERR_ARG; }
    if (B < 1 || B > c->cfg.max_batch) { set_error("batch %d exceeds max_batch %d", B, c->cfg.max_batch); return DYT_ERR_ARG; }
    if (!images || !trainable || !logits) { set_error("null argument"); return DYT_ERR_ARG; }
    if ((g1 == nullptr) != (g2 == nullptr)) { set_error("g1 and g2 must be given together"); return DYT_ERR_ARG; }
    const int P = c->prec, depth = c->cfg.depth, M = B * NT, r = c->cfg.ffn_num;
    const bool training = flags & DYT_F_TRAINING, complete = flags & DYT_F_COMPLETE, save = flags & DYT_F_SAVE;
    const bool masked_dense = (flags & DYT_F_MASKED_DENSE) && !complete;
    const bool dense = complete || masked_dense;
    const bool use_gate = !complete || (flags & DYT_F_GATE_ALWAYS);
    const float drop_p = training ? c->cfg.adapter_dropout : 0.f;
    const uint64_t* seed_dev = (flags & DYT_F_DEVICE_SEED) ? c->seed_dev : nullptr;
    static const int one_part_env = getenv("DYT_ONE_PART_COMPLETE") ? atoi(getenv("DYT_ONE_PART_COMPLETE")) : -1;   // measurement knob (bit per class: qkv 1, proj 2, fc1 4, fc2 8)
    const int one_part = (c->split16 && complete) ? (one_part_env >= 0 ? one_part_env : c->one_part_complete) : 0;
    const int fm = c->split16 ? (complete ? c->f8_mask_complete : c->f8_mask) : 0;   // classes (qkv 1, proj 2, fc1 4, fc2 8, embed 16) whose split operands are in the hi16 / fp8 form
    const bool planes = c->bwd16 && c->split16 && c->split_attn;   // q / k / v as 16-bit hi + lo planes (QKV epilogue -> split attention kernel; hi = what a 16-bit backward reads)
    const bool save16 = save && planes;   // "fp16x3h": what the backward reads is saved in the 16-bit operand type
    const bool fold = c->ln_fold && P != 0;   // LayerNorm-2 inside the fc1 GEMM (dyt_ctx::ln_fold)
    Slot& S = c->slots[slot];
    Transients& T = S.T;
    S.valid = false;
    if (B % c->frames != 0) { set_error("video model: batch %d is not a multiple of frames %d", B, c->frames); return DYT_ERR_ARG; }
    if (do_prep) { int rc = prep_adapters(c, trainable, s); if (rc) return rc; rc = prep_pool(c, trainable, s); if (rc) return rc; }
    hipStream_t sb = nullptr;
    { int rc = branch_stream(c, S, &sb); if (rc) return rc; }

    // stochastic depth (reference vision_transformer_IN21K.py:121,131,148,159; dpr = linspace(0, rate, depth), :285): training passes only;
    // every pass draws its own factors, like every call of the reference's forward does.  dp[(branch * depth + l) * B + b]
    const float* dp = nullptr;
    if (training && (S.dp_inject || c->drop_path_rate > 0.f)) {
        if (c->count_flops_tokens) { set_error("drop_path with the count_flops forward"); return DYT_ERR_STATE; }
        dp = S.dp_inject;
        if (!dp) {
            RUN(2, 0, launch_drop_path_draw(S.dp_own, depth, B, c->drop_path_rate, seed, seed_dev, ((uint64_t)slot << 32) | 0x10000ull, s));
            dp = S.dp_own;
        }
    }
    S.dp = dp;
    const bool tokens_in = flags & DYT_F_TOKENS_IN, tokens_out = flags & DYT_F_TOKENS_OUT;
    if (tokens_in) {
        // stand-alone Block.forward (reference vision_transformer_IN21K.py:144-165 called on a token tensor, as
        // block_flops_dict.py:36-46 does): `images` IS the residual stream [B,197,768]
        DYT_HIP_CHECK(hipMemcpyAsync(S.xs[0], images, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else if (!share0) {
        // patch embedding: im2col + GEMM (+bias +pos_embed), cls rows
        RUN(2, 0, launch_im2col(P, images, T.xn, B, s));
        {
            GemmArgs a; a.A = T.xn; a.W = c->pe_w; a.M = B * NP; a.N = D; a.K = D;
            a.bias = c->pe_b; a.pos = c->pos; a.out_f32 = S.xs[0]; SPLIT(a, c->pe_w3);
            if (fm & 16) { a.f8 = true; a.w_exp = c->pe_w_exp; }
            RUN_GEMM(EPI_EMBED, a);
        }
        RUN(2, 0, launch_cls_rows(c->cls, c->pos, S.xs[0], B, s));
    } else if (ev_b0_wait) {
        DYT_HIP_CHECK(hipStreamWaitEvent(s, ev_b0_wait, 0));  // the other pass's block-0 `u` is complete
    }

    for (int l = 0; l < depth; ++l) {
        const LayerW& W = c->W[l];
        LayerS& L = S.L[l];
        const float* base = trainable + (int64_t)l * c->layer_stride;
        float* x = S.xs[l];
        float* xo = S.xs[l + 1];
        // learnable scale: the up-projection copies / up_bp are already multiplied by it (prep_adapters_kernel)
        const float ad_scale = c->learn_scale ? 1.0f : c->cfg.adapter_scale;
        const float* up_bias = c->learn_scale ? c->ad_up_bp + (size_t)l * D : base + c->off_ub;
        const float* dp1 = (dp && l > 0) ? dp + (size_t)l * B : nullptr;             // attention branch (block 0: rate 0, never dropped)
        const float* dp2 = (dp && l > 0) ? dp + (size_t)(depth + l) * B : nullptr;   // MLP branch
        if (!(share0 && l == 0)) {
            RUN(2, 0, launch_ln_fwd(P, x, W.ln1_w, W.ln1_b, T.xn, L.st1, M, s, c->split16 ? T.xn3 : nullptr, fm & 1));
            {
                GemmArgs a; a.A = T.xn; a.W = W.qkv_w; a.Wp = W.qkv_wp; a.M = M; a.N = 3 * D; a.K = D; a.bias = W.qkv_b;
                a.out_at = L.q; a.out_at2 = L.k; a.out_at3 = L.v; SPLIT_F(a, W.qkv_w3, W.qkv_w3b, 0); SPLIT_READY(a, T.xn3);
                if (planes) { a.out_at = L.q16; a.out_at2 = L.k16; a.out_at3 = L.v16; a.qkv_lo[0] = T.qlo; a.qkv_lo[1] = T.klo; a.qkv_lo[2] = T.vlo; }
                RUN_GEMM(EPI_QKV, a);
            }
            void* ao3 = (c->split16 && c->split_attn && c->split_prod) ? ((save16 && L.ao3) ? L.ao3 : T.g3) : nullptr;   // the split attention kernel also writes the proj GEMM's operand
            // last block of a pass without a gate (teacher / complete model): the proj GEMM runs on the gathered cls rows of the fp32 output
            const bool tail_proj = c->cls_tail && l == depth - 1 && l > 0 && !tokens_out && !use_gate;
            AttnSave16 sv16{L.q16, L.k16, L.v16, nullptr};   // (the output's 16-bit copy is the hi plane of ao3)
            if (planes) { sv16.q_lo = T.qlo; sv16.k_lo = T.klo; sv16.v_lo = T.vlo; }   // bwd16: the 16-bit copies the backward reads (the fp32 output is then not needed once the proj operand is written)
            RUN(1, 4.0 * B * NH * (double)NT * NT * HD, launch_attn_fwd(P, L.q, L.k, L.v, (save16 && ao3 && !tail_proj) ? nullptr : L.attn_o, L.lse, B, s, c->split16 && c->split_attn, ao3, (save16 || planes) ? &sv16 : nullptr, (fm >> 1) & 1, (planes && (fm & 32)) ? 1 : 3));
            if (tail_proj) {
                // last block of a pass without a gate (teacher / complete model): only u[cls] is read downstream (LN2 / MLP / adapter of
                // the cls rows, their backward) -- the proj GEMM runs on the B gathered cls rows; same k order, same bits for those rows
                GemmArgs a; a.A = L.attn_o; a.a_map = c->cls_rows; a.W = W.proj_w; a.M = B; a.N = D; a.K = D; a.bias = W.proj_b;
                a.resid = x; a.out_f32 = L.u; a.scale = 1.0f; a.row_map = c->cls_rows; SPLIT_F(a, W.proj_w3, W.proj_w3b, 1);
                a.row_scale = dp1;
                RUN_GEMM(EPI_AD_UP, a);
            } else {
                GemmArgs a; a.A = L.attn_o; a.W = W.proj_w; a.Wp = W.proj_wp; a.M = M; a.N = D; a.K = D; a.bias = W.proj_b; a.resid = x;
                a.out_f32 = L.u; a.out_at = P == 0 ? nullptr : L.u_at; SPLIT_F(a, W.proj_w3, W.proj_w3b, 1);
                if (save16) { a.out_at = L.u16; a.save16 = true; }
                if (ao3) SPLIT_READY(a, ao3);
                if (fold) a.ln_part = L.ln_part;
                a.row_scale = dp1;
                RUN_GEMM(EPI_BIAS_RESID, a);
            }
            if (l == 0 && ev_b0_record) DYT_HIP_CHECK(hipEventRecord(ev_b0_record, s));
        }
        const bool tail = c->cls_tail && l == depth - 1 && !tokens_out;  // only the cls rows of the last block reach the head
        const int Mr = tail ? B : M;                       // rows the adapter / MLP of this block run on
        if (tail) {  // LN2 of the cls rows + their AT copy (adapter operand); everything below works on B rows
            RUN(2, 0, launch_ln_cls(P, L.u, W.ln2_w, W.ln2_b, T.xn, L.st2, S.ucls_at, B, s));
            if (save16) RUN(2, 0, launch_convert(1, (const float*)S.ucls_at, S.ucls16, (int64_t)B * D, s));
        }
        // ---- adapter branch: x_out = u + scale * up(dropout(relu(down(u)))) -- independent of the
        //      gate / gather / fc1 chain below, so it runs on the pass's side stream until fc2 needs x_out
        // 16-bit modes: wherever the MLP output h is not needed on its own (teacher pass, cls tail, inference) the
        // up-projection is the leading k-tile of the fc2 contraction (x_out = u + [d_act | h1] [s Wup | W2]^T + b): one read
        // of u and one write of x_out per row instead of two fp32 read-modify-write passes, and no up-projection launch.
        // In a compacted pass that covers the kept rows; the dropped rows get their u + adapter(u) from an
        // up-projection launch that skips the kept ones.  In a training student pass the saved MLP output h then includes the
        // adapter; the gate gradient <g, mlp(x)> is recovered in tok_bwd by subtracting <g, adapter(x)>, which the adapter's own
        // backward operands give for 128 B per token (TokBwdArgs::cat_*).  The masked mode keeps the two-launch form.
        const bool need_h = save && !complete && !tail;
        // the adapter's own LayerNorm (dyt_config::adapter_ln; reference models/dynamic_adapter.py:121-122 "in": down_proj reads LN_a(u); :132-133
        // "out": the scaled up-projection output goes through LN_a before it joins the residual stream).  Generic kernels, no fusion with fc2.
        const bool ad_in = c->ad_ln == 1, ad_out = c->ad_ln == 2;
        const float* aln_w = base + c->off_alw; const float* aln_b = base + c->off_alb;
        const bool cat = c->fc2_cat && P != 0 && !masked_dense && !dp2 && !ad_out;   // (a scaled MLP branch cannot share its accumulator with the adapter's)
        // Split fp32 forms whose backward runs on 16-bit operands ("fp16x3h", "fp16f8", "fp16x3q"; round 6): the same fusion with the up-projection as a
        // THREE-part product -- s d_act leaves the down-projection epilogue as a [hi | lo] image, W_up is split once per step (prep_adapters) --
        // contracted by the fc2 kernel as three leading tiles in front of its main loop (gemm.hip: LEAD); the dropped tokens' up-projection launch
        // runs on the same two images.  Until round 5 these modes ran the up-projection on the exact-fp32 MFMA kernel: an fp32
        // read-modify-write of [M,768] per block and pass (47 us) in front of the fc2 epilogue's own.
        static const bool cat3_env = !(getenv("DYT_FC2_CAT3") && atoi(getenv("DYT_FC2_CAT3")) == 0);   // measurement switch: 0 = the round-5 two-launch form in the split modes only
        const bool cat3 = cat3_env && !ad_out && c->fc2_cat && P == 0 && c->split16 && c->bwd16 && T.dact3 && c->ad_up_w3 && !masked_dense && !dp2;
        L.h_has_adapter = (cat || cat3) && need_h;   // the saved "MLP output" of this block then includes the adapter: tok_bwd corrects <g, h>
        FORK(sb);
        if (ad_in) RUN_ON(sb, 2, 0, launch_adapter_ln_fwd(P, L.u, aln_w, aln_b, T.xa, L.st_a, nullptr, Mr, s));
        {
            GemmArgs a; a.A = ad_in ? (const void*)T.xa : (tail ? S.ucls_at : L.u_at); a.W = at_off(c, c->ad_down_w, (size_t)l * RP * D); a.M = Mr; a.N = RP; a.K = D;
            a.bias = c->ad_down_b + l * RP; a.out_at = L.d_act; a.r = r; a.drop_p = drop_p;
            a.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
            a.keep = keep_mask ? keep_mask + (size_t)l * M * r : nullptr;
            a.row_map = tail ? c->cls_rows : nullptr;
            a.seed = seed; a.subseq = ((uint64_t)slot << 32) | (uint64_t)(l * 2 + 1); a.seed_dev = seed_dev;
            if (cat) { a.out_at2 = T.dact_s; a.scale = ad_scale; }
            if (save16) { a.out_at2 = L.dact16; a.scale = 1.0f; a.save16 = true; }
            if (cat3) { a.out3 = T.dact3; a.out3_scale = ad_scale; }
            RUN_ON(sb, 0, a.flops(), launch_gemm(P, EPI_AD_DOWN, a, s));
        }
        GemmArgs up; up.A = L.d_act; up.W = at_off(c, c->ad_up_w, (size_t)l * RP * D); up.M = Mr; up.N = D; up.K = RP;
        up.bias = up_bias; up.resid = L.u; up.out_f32 = xo; up.scale = ad_scale;
        up.row_map = tail ? c->cls_rows : nullptr;
        if (ad_out) {   // x_out = u + LN_a(s (d_act W_up^T + b_up)): the LayerNorm's input is kept (its backward needs x_hat), fc2 then adds in place
            GemmArgs g; g.A = L.d_act; g.W = at_off(c, c->ad_up_ws, (size_t)l * RP * D); g.M = Mr; g.N = D; g.K = RP;
            g.bias = c->ad_up_bp + (size_t)l * D; g.out_f32 = L.up32;
            RUN_ON(sb, 0, g.flops(), launch_gemm(P, EPI_BIAS_F32, g, s));
            RUN_ON(sb, 2, 0, launch_adapter_ln_fwd(0, L.up32, aln_w, aln_b, xo, L.st_a, L.u, Mr, s));
        } else if (!cat && !cat3) RUN_ON(sb, 0, up.flops(), launch_gemm(P, EPI_AD_UP, up, s));
        const uint16_t* up_w3 = cat3 ? (const uint16_t*)c->ad_up_w3 + (size_t)l * SPLIT_A * RP * D : nullptr;
        int* counts = S.counts + (size_t)l * B;
        if (use_gate) {
            GateArgs ga;
            ga.u = L.u; ga.w = base + c->off_gw; ga.b = base + c->off_gb;
            ga.g1 = g1 ? g1 + (size_t)l * B * NP : nullptr;
            ga.g2 = g2 ? g2 + (size_t)l * B * NP : nullptr;
            ga.batch = B; ga.training = training; ga.tau = c->cfg.tau; ga.threshold = c->cfg.threshold;
            ga.seed = seed; ga.subseq = ((uint64_t)slot << 32) | (uint64_t)(l * 2); ga.seed_dev = seed_dev;
            ga.soft = L.soft; ga.maskf = L.maskf;
            ga.out_select = token_select ? token_select + (size_t)l * NP : nullptr;
            ga.out_logits = token_logits ? token_logits + (size_t)l * NP : nullptr;
            ga.out_stride = depth * NP;
            ga.keep_local = L.keep_local; ga.counts = counts; ga.force_first = c->count_flops_tokens;
            RUN(2, 0, launch_gate(ga, s));
        }
        const bool fold2 = fold && !tail;   // LN2 inside fc1: (mean, rstd) from the proj epilogue's partials, no normalised copy of u
        if (fold2) {   // (the fc1 GEMM merges the partials itself)
            if (!dense || (masked_dense && save)) RUN(2, 0, launch_gather_index(L.keep_local, counts, L.total, L.maskf, L.row_src, L.dst_of, B, s, T.drop_src));
        } else if (tail) {
            // nothing: T.xn already holds LN2 of the cls rows
        } else if (!dense) {
            RUN(2, 0, launch_ln_gather(P, L.u, W.ln2_w, W.ln2_b, L.keep_local, counts, L.total, L.maskf, T.xn, L.st2,
                                       L.row_src, L.dst_of, B, s, c->split16 ? T.xn3 : nullptr, (fm >> 2) & 1, cat3 ? T.drop_src : nullptr));
        } else {
            RUN(2, 0, launch_ln_fwd(P, L.u, W.ln2_w, W.ln2_b, T.xn, L.st2, M, s, c->split16 ? T.xn3 : nullptr, (fm >> 2) & 1));
            // reference-style (masked) student pass: the MLP runs on every token, but its backward only has rows for the
            // kept ones (dH = mask * g) and is compacted -- it needs the dispatcher's index arrays too
            if (masked_dense && save) RUN(2, 0, launch_gather_index(L.keep_local, counts, L.total, L.maskf, L.row_src, L.dst_of, B, s));
        }
        // MLP on the kept (or all / cls) tokens, scatter-add into the residual stream
        const int* kdev = (dense || tail) ? nullptr : L.total;
        {
            GemmArgs a; a.A = T.xn; a.W = W.fc1_w; a.Wp = W.fc1_wp; a.M = Mr; a.N = DM; a.K = D; a.m_dev = kdev; a.bias = W.fc1_b;
            a.out_at = T.h1; a.out_at2 = save ? L.z : nullptr; SPLIT_F(a, W.fc1_w3, W.fc1_w3b, 2);
            if (save16) { a.out_at2 = L.z16; a.save16 = true; }
            if (!tail) SPLIT_READY(a, T.xn3);   // (the cls tail's LN2 rows come from ln_cls in fp32: pre-pass)
            if (c->split16) { a.out3 = T.h3; a.out3_f8 = (fm >> 3) & 1; }
            if (fold2) {
                a.A = L.u_at; a.a_map = dense ? nullptr : L.row_src; a.W = W.fc1_wf; a.Wp = W.fc1_wfp; a.bias = W.fc1_bf;
                a.ln_part = L.ln_part; a.ln_st_out = L.st2; a.ln_scratch = T.st_compact; a.ln_cs = W.fc1_cs;
            }
            RUN_GEMM(EPI_FC1, a);
        }
        JOIN(sb);  // x_out now holds u + adapter(u) (two-launch form) / d_act is complete
        if ((cat || cat3) && !dense && !tail) {   // dropped tokens: x_out = u + adapter(u) (the kept ones are written by fc2 below)
            if (cat3) {   // three-part on the [hi | lo] images; the operand carries the adapter scale, the bias takes it in the epilogue
                // (round 6b: over the list of dropped rows like the 16-bit modes' launch -- gathered operand rows, scattered output rows -- instead of
                // every row tile with the kept rows masked: 54 -> ~18 us)
                up.W3 = up_w3; up.a3 = T.dact3; up.a3_ready = true; up.scale = 1.0f; up.bias_scale = ad_scale;
                static const bool drop_list = !(getenv("DYT_CAT3_DROP_LIST") && atoi(getenv("DYT_CAT3_DROP_LIST")) == 0);
                if (drop_list && T.drop_src) { up.a3_mapped = true; up.a_map = T.drop_src; up.row_map = T.drop_src; up.m_dev = L.total + 1; }
                else up.row_mask = L.maskf;
            } else if (fold2) {   // over the dispatcher's list of dropped rows (gather + scatter) instead of every row with the kept ones skipped
                up.a_map = T.drop_src; up.row_map = T.drop_src; up.m_dev = L.total + 1;
            } else {
                up.row_mask = L.maskf;
            }
            RUN_GEMM(EPI_AD_UP, up);
        }
        {
            GemmArgs a; a.A = T.h1; a.W = W.fc2_w; a.M = Mr; a.N = D; a.K = DM; a.m_dev = kdev; a.bias = W.fc2_b;
            a.out_f32 = xo;
            a.row_map = tail ? c->cls_rows : (dense ? nullptr : L.row_src);
            a.row_mask = (masked_dense && !tail) ? L.maskf : nullptr;   // the cls token is never gated
            a.h_out = need_h ? L.h : nullptr;                           // cls rows carry no gate gradient
            if (save16 && need_h) { a.h_out = L.h16; a.save16 = true; }
            SPLIT_F(a, W.fc2_w3, W.fc2_w3b, 3); SPLIT_READY(a, T.h3);
            if (cat3) {
                a.A2 = T.dact3; a.W2 = up_w3;
                a.a2_map = (dense || tail) ? nullptr : L.row_src;
                a.bias2 = up_bias; a.scale = ad_scale; a.resid = L.u;
            }
            if (cat) {
                a.A2 = T.dact_s; a.W2 = at_off(c, c->ad_up_w, (size_t)l * RP * D);   // [s d_act | h] x [W_up | W2]^T
                a.a2_map = (dense || tail) ? nullptr : L.row_src;   // d_act is indexed by token (cls tail: by image, like h1)
                a.bias2 = up_bias; a.scale = ad_scale; a.resid = L.u;
            }
            a.row_scale = dp2;
            if (tail) { a.splitk_ws = (float*)T.dZ; a.splitk_ws_bytes = (size_t)M * DM * c->at; }   // (a backward-pass buffer: idle here)
            RUN_GEMM(EPI_FC2, a);
        }
    }
    if (tokens_out) {   // the block stack's output tokens instead of the head: `logits` receives [B,197,768]
        DYT_HIP_CHECK(hipMemcpyAsync(logits, S.xs[depth], (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else if (c->frames > 1) {
        int rc = pool_forward(c, S, trainable, logits, B, s);
        if (rc) return rc;
    } else {
        RUN(2, 0, launch_head_fwd(S.xs[depth], c->norm_w, c->norm_b, trainable + c->off_hw, trainable + c->off_hb, S.cls_n,
                                  S.head_stats, logits, B, c->cfg.num_classes, s));
    }
    c->pass_ran = true;
    S.batch = B; S.flags = flags; S.valid = save && !tokens_in && !tokens_out; S.trainable = trainable; S.saved16 = save16;   // token-level passes are forward only
    return DYT_OK;
}

extern "C" int dyt_forward(dyt_ctx* c, int slot, const float* images, int batch, int flags, const float* trainable,
                           const float* g1, const float* g2, const uint8_t* keep_mask, uint64_t seed, float* logits,
                           float* token_select, float* token_logits, void* stream) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    if (slot >= 0 && slot < c->cfg.slots) {  // a stand-alone pass owns its block-0 buffers
        Slot& S = c->slots[slot];
        S.L[0].u = S.u0_own; S.L[0].u_at = S.u0_at_own; S.L[0].u16 = S.u0_16_own; S.L[0].ln_part = S.part0_own;
    }
    return forward_impl(c, slot, images, batch, flags, trainable, g1, g2, keep_mask, seed, logits, token_select,
                        token_logits, true, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// Measurement hook (tools/probes/determinism_trace.py): DYT_DBG_CKSUM=1 -> after every backward launch an order-independent
// integer checksum of its output buffer is taken on the same stream into a per-pass log; two runs of the same step are then
// compared launch by launch: the first differing entry names the kernel whose output is not reproducible.
__global__ void dbg_checksum_kernel(const uint32_t* __restrict__ p, size_t nwords, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256)
        acc += (unsigned long long)p[i] * (unsigned long long)((i & 0xffff) + 1);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
struct DbgCk { unsigned long long* dev = nullptr; int n[2] = {0, 0}; int on = -1; std::vector<std::string> label[2];
               void* dump = nullptr; size_t dump_bytes = 0, dump_len = 0; };
static DbgCk g_ck;
constexpr int DBG_CK_MAX = 1024;
static bool dbg_ck_on() {
    if (g_ck.on < 0) { const char* e = getenv("DYT_DBG_CKSUM"); g_ck.on = e && atoi(e) ? 1 : 0; }
    return g_ck.on == 1;
}
static int dbg_ck_reset(hipStream_t s) {
    if (!dbg_ck_on()) return 0;
    if (!g_ck.dev) DYT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g_ck.dev), 2 * DBG_CK_MAX * sizeof(unsigned long long)));
    DYT_HIP_CHECK(hipMemsetAsync(g_ck.dev, 0, 2 * DBG_CK_MAX * sizeof(unsigned long long), s));
    for (int i = 0; i < 2; ++i) { g_ck.n[i] = 0; g_ck.label[i].clear(); }
    return 0;
}
static int dbg_ck(int slot, hipStream_t s, const char* what, int layer, const void* p, size_t bytes) {
    if (!dbg_ck_on() || !p || slot > 1 || g_ck.n[slot] >= DBG_CK_MAX) return 0;
    char buf[64];
    snprintf(buf, sizeof(buf), "L%d %s", layer, what);
    static const char* only = getenv("DYT_DBG_CKSUM_ONLY");   // comma-separated substrings: trace only the matching launches
    if (only && only[0]) {
        bool hit = false;
        std::string pats(only);
        for (size_t a = 0; a <= pats.size();) {
            const size_t e = pats.find(',', a) == std::string::npos ? pats.size() : pats.find(',', a);
            if (e > a && strstr(buf, pats.substr(a, e - a).c_str())) hit = true;
            a = e + 1;
        }
        if (!hit) return 0;
    }
    g_ck.label[slot].push_back(buf);
    static const char* dump = getenv("DYT_DBG_DUMP");   // "<slot>:<label>": keep a copy of that launch's output buffer
    if (dump && dump[0] && dump[1] == ':' && dump[0] - '0' == slot && !strcmp(dump + 2, buf)) {
        if (g_ck.dump_bytes < bytes) {
            if (g_ck.dump) (void)hipFree(g_ck.dump);
            DYT_HIP_CHECK(hipMalloc(&g_ck.dump, bytes));
            g_ck.dump_bytes = bytes;
        }
        DYT_HIP_CHECK(hipMemcpyAsync(g_ck.dump, p, bytes, hipMemcpyDeviceToDevice, s));
        g_ck.dump_len = bytes;
    }
    hipLaunchKernelGGL(dbg_checksum_kernel, dim3(128), dim3(256), 0, s, static_cast<const uint32_t*>(p), bytes / 4,
                       g_ck.dev + slot * DBG_CK_MAX + g_ck.n[slot]++);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}
#define CK(what, ptr, bytes) do { int _r = dbg_ck(slot, s, (what), l, (ptr), (bytes)); if (_r) return _r; } while (0)
extern "C" int dyt_debug_checksums(int slot, uint64_t* out, int max_n, int* n_out) {
    if (!out || !n_out || slot < 0 || slot > 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    *n_out = 0;
    if (!g_ck.dev) return DYT_OK;
    DYT_HIP_CHECK(hipDeviceSynchronize());
    const int n = g_ck.n[slot] < max_n ? g_ck.n[slot] : max_n;
    DYT_HIP_CHECK(hipMemcpy(out, g_ck.dev + slot * DBG_CK_MAX, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *n_out = n;
    return DYT_OK;
}
extern "C" int64_t dyt_debug_dump_read(void* dst_device, int64_t max_bytes) {   // -> bytes copied (device to device)
    if (!g_ck.dump || !dst_device) return 0;
    const size_t n = g_ck.dump_len < (size_t)max_bytes ? g_ck.dump_len : (size_t)max_bytes;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(dst_device, g_ck.dump, n, hipMemcpyDeviceToDevice) != hipSuccess) return -1;
    return (int64_t)n;
}
extern "C" const char* dyt_debug_checksum_label(int slot, int i) {
    if (slot < 0 || slot > 1 || i < 0 || i >= (int)g_ck.label[slot].size()) return "";
    return g_ck.label[slot][i].c_str();
}

// Measurement hook (tools/probes/determinism_cumask.py, PMASK=iso): DYT_DBG_ISO = bit mask of backward kernel classes that are
// launched on a per-pass stream pinned to ONE shader engine of every XCD (mask bits 24..31 of each word), fork/joined with
// events around every launch; the probe pins the two pass streams to the other three engines.  Finds which kernel class
// must share CUs with the other pass for the run-to-run differences of DESIGN.md 7b to appear.
//   1 attention bwd   2 tok_bwd (+ its reduce)   4 ln_bwd   8 frozen-weight dgrad GEMMs   16 adapter dgrad GEMMs   32 wgrad   64 prep / head
struct DbgIso { hipStream_t st[4] = {}; hipEvent_t a[4] = {}, b[4] = {}; int mask = -1; };
static DbgIso g_iso;
static int dbg_iso_mask() {
    if (g_iso.mask < 0) { const char* e = getenv("DYT_DBG_ISO"); g_iso.mask = e ? atoi(e) : 0; }
    return g_iso.mask;
}
static int dbg_iso_enter(int slot, int cls, hipStream_t* s, hipStream_t* keep) {
    *keep = *s;
    if (!(dbg_iso_mask() & cls)) return 0;
    if (!g_iso.st[slot]) {
        uint32_t words[8];
        for (int i = 0; i < 8; ++i) words[i] = 0xFF000000u;
        DYT_HIP_CHECK(hipExtStreamCreateWithCUMask(&g_iso.st[slot], 8, words));
        DYT_HIP_CHECK(hipEventCreateWithFlags(&g_iso.a[slot], hipEventDisableTiming));
        DYT_HIP_CHECK(hipEventCreateWithFlags(&g_iso.b[slot], hipEventDisableTiming));
    }
    DYT_HIP_CHECK(hipEventRecord(g_iso.a[slot], *s));
    DYT_HIP_CHECK(hipStreamWaitEvent(g_iso.st[slot], g_iso.a[slot], 0));
    *s = g_iso.st[slot];
    return 0;
}
static int dbg_iso_leave(int slot, hipStream_t* s, hipStream_t keep) {
    if (*s == keep) return 0;
    DYT_HIP_CHECK(hipEventRecord(g_iso.b[slot], *s));
    *s = keep;
    DYT_HIP_CHECK(hipStreamWaitEvent(*s, g_iso.b[slot], 0));
    return 0;
}
#define ISO(cls, body)                                                        \
    do {                                                                      \
        hipStream_t _iso_keep;                                                \
        { int _r = dbg_iso_enter(slot, (cls), &s, &_iso_keep); if (_r) return _r; } \
        body                                                                  \
        { int _r = dbg_iso_leave(slot, &s, _iso_keep); if (_r) return _r; }   \
    } while (0)

// Measurement hook: DYT_DBG_POISON = bit mask of backward transients that are filled with NaN bit patterns (0xFF bytes) on the
// stream right before the kernel that produces them.  A consumer that reads a row before its producer's store is visible then
// turns the final gradient into NaN instead of a 1e-6 difference.   1 dxn  2 dA2  4 dad  8 du_at  16 dO  32 dqkv  64 ddz  128 dZ  256 g_at
static int dbg_poison(int bit, void* p, size_t bytes, hipStream_t s) {
#ifndef DYT_DEBUG_HOOKS
    (void)bit; (void)p; (void)bytes; (void)s;
    return 0;   // measurement builds only (-DDYT_DEBUG_HOOKS)
#endif
    static int mask = -1;
    if (mask < 0) { const char* e = getenv("DYT_DBG_POISON"); mask = e ? atoi(e) : 0; }
    if (!(mask & bit) || !p) return 0;
    DYT_HIP_CHECK(hipMemsetAsync(p, 0xFF, bytes, s));
    return 0;
}
#define POISON(bit, ptr, bytes) do { int _r = dbg_poison((bit), (ptr), (bytes), s); if (_r) return _r; } while (0)

// ev_split (optional) is recorded on `s` once the gradients of the head and of every block >= split are enqueued
static int backward_impl(dyt_ctx* c, int slot, const float* trainable, const float* dlogits, const float* dtoken_select,
                         const float* dtok, const float* dtoken_logits, float* grad, hipStream_t s,
                         hipEvent_t ev_split = nullptr, int split = 0) {
    if (slot < 0 || slot >= c->cfg.slots) { set_error("slot %d out of range", slot); return DYT_ERR_ARG; }
    Slot& S = c->slots[slot];
    Transients& T = S.T;
    if (!S.valid) { set_error("slot %d holds no saved forward (call dyt_forward with DYT_F_SAVE)", slot); return DYT_ERR_STATE; }
    if (!dlogits || !grad || !trainable) { set_error("null argument"); return DYT_ERR_ARG; }
    // b16 ("fp16x3h"): the saved pass came from the exact (split fp32) forward, this backward runs in the 16-bit mode: P = 1, the
    // 16-bit copies of the saved tensors and of the dgrad matrices, none of the split forms below
    const bool b16 = S.saved16;
    const int P = b16 ? 1 : c->prec, depth = c->cfg.depth, B = S.batch, M = B * NT, r = c->cfg.ffn_num;
    const bool split16 = c->split16 && !b16;
    const size_t atb = at_size(P);
    auto at_offb = [atb](void* base, size_t elems) { return static_cast<void*>(static_cast<char*>(base) + elems * atb); };
    const int flags = S.flags;
    const bool training = flags & DYT_F_TRAINING, complete = flags & DYT_F_COMPLETE;
    // masked_dense: the student forward evaluated the MLP for every token and multiplied by the mask (the reference's
    // training semantics: h and gelu' exist for all tokens, indexed by token row).  Its BACKWARD is compacted all the same:
    // the rows of dH = mask * g that belong to dropped tokens are exactly zero, so dZ / dA2 are computed for the kept rows
    // only (gathered through row_src) -- "exact-gradient" mode at 133.5 instead of 139.6 GFLOP per image.
    const bool masked_dense = (flags & DYT_F_MASKED_DENSE) && !complete;
    const bool dense = complete;               // MLP backward over all rows (teacher pass)
    const bool h_by_token = masked_dense;      // saved h / gelu' are indexed by token row, not by compact row
    const bool student = !complete;
    const float drop_p = training ? c->cfg.adapter_dropout : 0.f;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const float scale = c->learn_scale ? 1.0f : c->cfg.adapter_scale;   // (learnable: the dgrad matrices carry it, the weight gradients get it in learn_scale_fixup_kernel)
    if (c->learn_scale) DYT_HIP_CHECK(hipMemsetAsync(S.gscr, 0, (size_t)depth * c->layer_stride * sizeof(float), s));
    const float gs = P == 0 ? 1.0f : c->gs, inv_gs = 1.0f / gs;   // 16-bit gradient operands carry gs (dyt_ctx: gs)
    float* g = T.g;
    // ... in the split modes' 16-bit backward as well (fp16x3q: 35.2 -> 34.5 ms per step same-box; worst gradient over the five seeds 1.40e-3 -> 1.53e-3, typical
    // 7e-4 -> 1.1e-3: tests/test_gpu_round4.py prints the table); DYT_G16_B16=0 keeps the fp32 stream there
    static const bool g16_b16 = !(getenv("DYT_G16_B16") && atoi(getenv("DYT_G16_B16")) == 0);
    const bool g16 = c->g16 && P == 1 && (!b16 || g16_b16);   // the gradient stream between the row kernels as 16-bit operand copies only (dyt_ctx::g16)
    hipStream_t sb = nullptr;
    { int rc = branch_stream(c, S, &sb); if (rc) return rc; }

    const bool cls_tail = c->cls_tail;
    if (c->frames > 1) {
        int rc = pool_backward(c, S, trainable, dlogits, grad, s);
        if (rc) return rc;
    } else {
        RUN(2, 0, launch_head_bwd(dlogits, S.xs[depth], S.cls_n, S.head_stats, c->norm_w, trainable + c->off_hw,
                                  cls_tail ? S.gcls : g, grad + c->off_hw, grad + c->off_hb, B, c->cfg.num_classes,
                                  cls_tail ? 1 : 0, s));
        { const int l = depth; CK("head_bwd g", cls_tail ? S.gcls : g, (size_t)(cls_tail ? B : M) * D * 4); CK("head_bwd dW", grad + c->off_hw, (size_t)c->cfg.num_classes * D * 4); }
    }

    int fix_hi = depth;   // learnable scale: blocks [0, fix_hi) still await learn_scale_fixup_kernel
    ReduceQueue rq;   // adapter weight-gradient / gate-gradient reductions: queued per block, flushed where the gradients must be final
    bool prepped = false;  // the previous iteration's ln_bwd already produced g_at / dmask for this block
    bool g3_ready = false; // ... and (fp32 split form) T.g3 = g as the 16-bit split operand of this block's GELU' dgrad
    const bool split_prod = split16 && c->split_prod;
    for (int l = depth - 1; l >= 0; --l) {
        const LayerW& W = c->W[l];
        LayerS& L = S.L[l];
        const float* base = trainable + (int64_t)l * c->layer_stride;
        float* gbase = grad + (int64_t)l * c->layer_stride;
        float* ubase = c->learn_scale ? S.gscr + (int64_t)l * c->layer_stride : gbase;   // where the up-projection's weight / bias gradients go
        const bool first = l == 0;
        // saved tensors / dgrad matrices in this backward's operand type
        const void* Lh = b16 ? L.h16 : L.h; const void* Ldact = b16 ? L.dact16 : L.d_act; const void* Lz = b16 ? L.z16 : L.z;
        const void* Luat = b16 ? L.u16 : L.u_at; const void* ucls = b16 ? S.ucls16 : S.ucls_at;
        const void *fc2_wT = b16 ? W.fc2_wT16 : W.fc2_wT, *fc2_wTp = b16 ? W.fc2_wTp16 : W.fc2_wTp, *fc1_wT = b16 ? W.fc1_wT16 : W.fc1_wT,
                   *fc1_wTp = b16 ? W.fc1_wTp16 : W.fc1_wTp, *proj_wT = b16 ? W.proj_wT16 : W.proj_wT, *proj_wTp = b16 ? W.proj_wTp16 : W.proj_wTp,
                   *qkv_wT = b16 ? W.qkv_wT16 : W.qkv_wT, *qkv_wTp = b16 ? W.qkv_wTp16 : W.qkv_wTp;
        void* ad_up_wT = b16 ? c->ad_up_wT16 : c->ad_up_wT; void* ad_down_wT = b16 ? c->ad_down_wT16 : c->ad_down_wT;
        void* Tdad = b16 ? T.dad16 : T.dad;

        const bool tail = cls_tail && l == depth - 1;  // incoming gradient lives at the cls rows only (S.gcls)
        const int Mr = tail ? B : M;
        float* gin = tail ? S.gcls : g;
        // ---- 1. prep: AT copy of g, gathered/masked MLP gradient rows, <g,h> per token ----
        void* g_at = P == 0 ? nullptr : T.g_at;
        if (!prepped && (g_at || (student && !tail))) {
            BwdPrepArgs a;
            a.g = gin; a.h = (student && !tail) ? Lh : nullptr; a.dst_of = (dense || tail || h_by_token) ? nullptr : L.dst_of;
            a.row_mask = nullptr;
            a.g_at = g_at; a.dH = nullptr; a.dmask = (student && !tail) ? T.dmask : nullptr;
            a.M = Mr; a.gs = gs;
            ISO(64, RUN(2, 0, launch_bwd_prep(P, a, s)););
            if (g_at) CK("bwd_prep g_at", g_at, (size_t)Mr * D * atb);
            if (a.dmask) CK("bwd_prep dmask", T.dmask, (size_t)Mr * 4);
        }
        const void* A_g = g_at ? g_at : (const void*)gin;
        const int* kdev = (dense || tail) ? nullptr : L.total;
        // the adapter's own LayerNorm (dyt_config::adapter_ln).  "out": the adapter branch sees the gradient BEHIND the LayerNorm -- its parameter
        // gradients (dy = g, x_hat from the saved input) and dup = LNbwd(g), which replaces g as the operand of the up-projection's dgrad / wgrad.
        const bool ad_in = c->ad_ln == 1, ad_out = c->ad_ln == 2;
        const void* A_ad = A_g;
        if (ad_out) {
            RUN(2, 0, launch_ln_param_grad(P, A_g, L.up32, L.st_a, T.aln_part, gbase + c->off_alw, Mr, gs, s));
            RUN(2, 0, launch_ln_bwd(P, A_g, L.up32, L.st_a, base + c->off_alw, nullptr, P == 0 ? T.dup : nullptr, Mr, P != 0 ? (void*)T.dup : nullptr,
                                    nullptr, nullptr, nullptr, gs, s));
            A_ad = T.dup;
        }
        if (ad_in) RUN(2, 0, launch_adapter_ln_fwd(P, L.u, base + c->off_alw, base + c->off_alb, T.xa, nullptr, nullptr, Mr, s));   // down_proj's wgrad operand LN_a(u), recomputed
        // ---- 2. adapter branch on the side stream: dgrad through up_proj, both wgrads, bias grads ----
        FORK(sb);
        {
            GemmArgs a; a.A = A_ad; a.W = at_offb(ad_up_wT, (size_t)l * RP * D); a.M = Mr; a.N = RP; a.K = D;
            a.aux_at = Ldact; a.out_at = T.ddz; a.scale = scale; a.inv_keep = inv_keep;
            POISON(64, T.ddz, (size_t)Mr * RP * atb);
            ISO(16, RUN_ON(sb, 0, a.flops(), launch_gemm(P, EPI_AD_DGRAD_UP, a, s)););
            CK("ad_dgrad_up ddz", T.ddz, (size_t)Mr * RP * atb);
        }
        {   // both weight gradients (+ the two bias gradients as ones columns / rows) in one launch
            WgradArgs w[2];
            WgradArgs& a = w[0];
            a.X = A_ad; a.Y = Ldact; a.M = Mr; a.r = r; a.partial = S.wg_part[l];
            a.out_w = ubase + c->off_uw; a.sc = r; a.sj = 1; a.alpha = scale * inv_gs;       // up_proj.weight [768, r]  (X = g_at carries gs)
            a.out_xsum = ubase + c->off_ub; a.alpha_x = scale * inv_gs;                      // up_proj.bias
            WgradArgs& b = w[1];
            b.X = ad_in ? (const void*)T.xa : (tail ? ucls : Luat); b.Y = T.ddz; b.M = Mr; b.r = r; b.partial = S.wg_part2[l];
            b.out_w = gbase + c->off_dw; b.sc = 1; b.sj = D; b.alpha = inv_gs;      // down_proj.weight [r, 768]  (Y = ddz carries gs)
            b.out_xsum = nullptr; b.alpha_x = 0.f;
            b.out_ysum = gbase + c->off_db; b.alpha_y = inv_gs;                     // down_proj.bias
            if (split16 && c->split_bwd_parts == 1 && c->split_wgrad16) {   // "fp16x3f": gradient products one-part here too
                a.half_products = b.half_products = true;
                a.x_scale = c->split_gs;   // X = g (gradient-sized), Y = d_act
                b.y_scale = c->split_gs;   // X = u, Y = ddz (gradient-sized)
            }
            ISO(32, RUN_ON(sb, 2, 4.0 * Mr * D * (double)RP, launch_wgrad(P, w, 2, s, sb ? nullptr : &rq)););
            CK("wgrad up_w", gbase + c->off_uw, (size_t)D * r * 4); CK("wgrad down_w", gbase + c->off_dw, (size_t)D * r * 4);
        }
        // ---- 3. MLP dgrad (frozen weights) on the main stream: dZ = (dH W2) * gelu'(z) ; dA2 = dZ W1 ----
        if (!first) {
            {
                GemmArgs a; a.A = A_g; a.W = fc2_wT; a.Wp = fc2_wTp; a.M = Mr; a.N = DM; a.K = D; a.m_dev = kdev; a.aux_at = Lz;
                a.a_map = (dense || tail) ? nullptr : L.row_src; a.out_at = T.dZ;   // kept rows of g (mask = 1 there) gathered by the loader
                a.row_map = (h_by_token && !tail) ? L.row_src : nullptr; if (split16) SPLIT_G(a, W.fc2_wT3);
                if (g3_ready && !tail) { SPLIT_READY(a, T.g3); a.a3_mapped = true; }   // ln_bwd of the block above wrote g as the split operand
                if (split16) { a.out3 = T.h3; a.out3_scale = c->split_gs; a.out3_hi_only = c->split_bwd_parts == 1; }   // dZ as the split operand of the fc1 dgrad
                if (dense) POISON(128, T.dZ, (size_t)Mr * DM * atb);
                ISO(8, RUN_GEMM(EPI_GELU_BWD, a););
                CK("gelu_bwd dZ", T.dZ, (size_t)Mr * DM * atb);
            }
            {
                GemmArgs a; a.A = T.dZ; a.W = fc1_wT; a.Wp = fc1_wTp; a.M = Mr; a.N = D; a.K = DM; a.m_dev = kdev; a.out_at = T.dA2; if (split16) { SPLIT_G(a, W.fc1_wT3); SPLIT_READY(a, T.h3); }
                if (tail) { a.splitk_ws = (float*)T.dqkv; a.splitk_ws_bytes = (size_t)M * 3 * D * atb; }   // (written by this block's attention backward, later)
                if (dense) POISON(2, T.dA2, (size_t)Mr * D * atb);
                ISO(8, RUN_GEMM(EPI_STORE_AT, a););
                CK("fc1_dgrad dA2", T.dA2, (size_t)Mr * D * atb);
            }
        }
        JOIN(sb);
        // adapter dgrad ddz Wdown.  fp32 mode / cls tail: accumulated into g in place (fp32 read-modify-write of [M,768]);
        // bf16 mode: stored as a bf16 [M,768] operand that tok_bwd adds (half the bytes of the in-place update)
        const bool dad_at = P != 0 && !tail && !first;
        if (ad_in) {
            // "in": ddz W_down is the gradient w.r.t. LN_a(u): the LayerNorm's parameter gradients from it (block 0 included), then its input gradient --
            // 16-bit backward: a 16-bit operand tok_bwd adds (T.dup); fp32 backward: accumulated into g
            GemmArgs a; a.A = T.ddz; a.W = at_offb(ad_down_wT, (size_t)l * RP * D); a.M = Mr; a.N = D; a.K = RP;
            if (P != 0) { a.out_at = Tdad; ISO(16, RUN_GEMM(EPI_STORE_AT, a);); }
            else { a.out_f32 = T.dup; a.accumulate = 0; a.scale = 1.0f; ISO(16, RUN_GEMM(EPI_STORE_F32, a);); }
            const void* dln = P != 0 ? (const void*)Tdad : (const void*)T.dup;
            RUN(2, 0, launch_ln_param_grad(P, dln, L.u, L.st_a, T.aln_part, gbase + c->off_alw, Mr, gs, s));
            if (!first) {
                if (P != 0) RUN(2, 0, launch_ln_bwd(P, dln, L.u, L.st_a, base + c->off_alw, nullptr, nullptr, Mr, T.dup, nullptr, nullptr, nullptr, gs, s));
                else RUN(2, 0, launch_ln_bwd(P, dln, L.u, L.st_a, base + c->off_alw, gin, gin, Mr, nullptr, nullptr, nullptr, nullptr, gs, s));
            }
        } else if (!first) {
            GemmArgs a; a.A = T.ddz; a.W = at_offb(ad_down_wT, (size_t)l * RP * D); a.M = Mr; a.N = D; a.K = RP;
            if (dad_at) { a.out_at = Tdad; POISON(4, Tdad, (size_t)Mr * D * atb); ISO(16, RUN_GEMM(EPI_STORE_AT, a);); }
            else { a.out_f32 = gin; a.accumulate = 1; a.scale = inv_gs; ISO(16, RUN_GEMM(EPI_STORE_F32, a);); }
            if (dad_at) CK("ad_dgrad_down dad", Tdad, (size_t)Mr * D * atb);   // g <- g + ddz Wdown
        }

        // ---- 4. per-token tail: LN2 backward scattered back, gate backward, AT copy of dL/du ----
        if (!first || student) {
            TokBwdArgs a;
            a.du = g; a.dA2 = first ? nullptr : T.dA2;
            a.dst_of = (dense || tail) ? nullptr : L.dst_of; a.u = L.u; a.stats2 = L.st2;
            if (g16 && !first) { a.du = nullptr; a.du_in_at = tail ? nullptr : T.g_at; }   // in: ln_bwd's (or bwd_prep's) 16-bit copy; out: du_at only
            a.ln2_w = W.ln2_w; a.gate_w = student ? base + c->off_gw : nullptr; a.soft = L.soft; a.maskf = L.maskf;
            a.dmask = tail ? nullptr : T.dmask;
            a.g_cls = tail ? S.gcls : nullptr;
            a.dad = dad_at ? (ad_in ? (void*)T.dup : Tdad) : nullptr;
            a.gs = gs; a.inv_gs = inv_gs;
            if (L.h_has_adapter && student && !tail) {
                a.cat_dact = Ldact; a.cat_ddz = T.ddz; a.cat_bup = base + c->off_ub;
                a.cat_scale = scale; a.cat_ddz_scale = 1.0f / (inv_keep * gs);
            }
            a.dtoken_select = dtoken_select ? dtoken_select + (size_t)l * NP : nullptr;
            a.dtoken_logits = dtoken_logits ? dtoken_logits + (size_t)l * NP : nullptr;
            a.dtok = dtok; a.out_stride = depth * NP; a.training = training; a.tau = c->cfg.tau;
            a.du_at = (P != 0 && !first) ? T.du_at : nullptr; a.partial = S.tok_part[l]; a.M = M; a.write_du = !first;
            a.branch_scale = (S.dp && l > 0) ? S.dp + (size_t)(depth + l) * B : nullptr;
            if (split_prod && !first) { a.du3 = T.g3; a.du3_scale = c->split_gs; a.du3_hi_only = c->split_bwd_parts == 1; }
            int nblk = 0;
            if (a.du_at) POISON(8, T.du_at, (size_t)M * D * atb);
            ISO(2, RUN(2, 0, launch_tok_bwd(P, a, &nblk, s)););
            if (student && !dbg_skip(8)) { int _r = queue_tok_reduce(rq, S.tok_part[l], nblk, gbase + c->off_gw); if (_r) return _r; }
            if (a.write_du && a.du) CK("tok_bwd g", g, (size_t)M * D * 4);
            if (a.dmask) CK("tok_in dmask", T.dmask, (size_t)M * 4);          // what tok_bwd consumed (unchanged by it)
            if (a.dA2) CK("tok_in dA2", T.dA2, (size_t)Mr * D * atb);
            if (a.dad) CK("tok_in dad", Tdad, (size_t)Mr * D * atb);
            if (a.du_at) CK("tok_bwd du_at", T.du_at, (size_t)M * D * atb);
            if (student) CK("tok_bwd gate grad", gbase + c->off_gw, (size_t)(D + 1) * 4);
        }
        if (l == split || first || dbg_ck_on()) RUN(2, 0, flush_reductions(rq, s));   // the gradients of blocks >= l are final after this
        if (c->learn_scale && ev_split && l == split && l > 0) {   // the upper blocks' up-projection / scale gradients must be final before the event as well
            hipLaunchKernelGGL(learn_scale_fixup_kernel, dim3(fix_hi - l), dim3(256), 0, s, S.gscr, trainable, grad, c->layer_stride, c->off_uw, c->off_ub, c->off_sc, r, l);
            DYT_HIP_CHECK(hipGetLastError());
            fix_hi = l;
        }
        if (ev_split && l == split) {
            // video model: the pooling head's k / v weight gradients (side stream, part 0 of the flat buffer) must be final
            // before the "upper gradients are final" event that the gradient sum and the early all-reduce wait for
            if (S.pool.wpending) { DYT_HIP_CHECK(hipStreamWaitEvent(s, S.pool.ev_wj, 0)); S.pool.wpending = false; }
            DYT_HIP_CHECK(hipEventRecord(ev_split, s));
        }
        if (first) break;
        // ---- 5. attention branch: proj dgrad, attention backward, qkv dgrad, LN1 backward ----
        {
            GemmArgs a; a.A = P == 0 ? (const void*)g : (const void*)T.du_at; a.W = proj_wT; a.Wp = proj_wTp; a.M = M; a.N = D; a.K = D;
            a.out_at = T.dO; if (split16) SPLIT_G(a, W.proj_wT3); if (split_prod) SPLIT_READY(a, T.g3);
            POISON(16, T.dO, (size_t)M * D * atb);
            ISO(8, RUN_GEMM(EPI_STORE_AT, a););
            if (S.dp) RUN(2, 0, launch_scale_rows(P, T.dO, S.dp + (size_t)l * B, M, D, s));   // stochastic depth: the attention branch's factor
            CK("proj_dgrad dO", T.dO, (size_t)M * D * atb);
        }
        POISON(32, T.dqkv, (size_t)M * 3 * D * atb);
        ISO(1, RUN(1, 14.0 * B * NH * (double)NT * NT * HD,
            launch_attn_bwd(P, b16 ? L.q16 : L.q, b16 ? L.k16 : L.k, b16 ? L.v16 : L.v, b16 ? L.ao3 : L.attn_o, T.dO, L.lse, T.delta, T.dqkv, B, s, (tail && !student) ? 1 : 7,
                            split16 ? T.dqkv3 : nullptr, c->split_gs, split16 && c->split_attn, c->split_bwd_attn_parts, c->split_bwd_parts == 1, b16 ? SPLIT_A * D : 0)););   // teacher tail: du, hence dO, is zero off the cls rows
        CK("attn_bwd delta", T.delta, (size_t)B * NH * NT * 4); CK("attn_bwd dqkv", T.dqkv, (size_t)M * 3 * D * atb);
        {
            GemmArgs a; a.A = T.dqkv; a.W = qkv_wT; a.Wp = qkv_wTp; a.M = M; a.N = D; a.K = 3 * D; a.out_at = T.dxn; if (split16) { SPLIT_G(a, W.qkv_wT3); SPLIT_READY(a, T.dqkv3); }
            POISON(1, T.dxn, (size_t)M * D * atb);
            ISO(8, RUN_GEMM(EPI_STORE_AT, a););
            CK("qkv_dgrad dxn", T.dxn, (size_t)M * D * atb);
        }
        {   // LN1 backward, fused with the next block's prep (AT copy of g, <g, h> for the gate gradient)
            const LayerS& Ln = S.L[l - 1];
            if (g_at) POISON(256, g_at, (size_t)M * D * atb);
            ISO(4, RUN(2, 0, launch_ln_bwd(P, T.dxn, S.xs[l], L.st1, W.ln1_w, g16 ? nullptr : g, g16 ? nullptr : g, M, g_at, student ? (b16 ? Ln.h16 : Ln.h) : nullptr,
                                    (!dense && !h_by_token) ? Ln.dst_of : nullptr, student ? T.dmask : nullptr, gs, s,
                                    (split_prod && l > 1) ? T.g3 : nullptr, c->split_gs, c->split_bwd_parts == 1, g16 ? T.du_at : nullptr)););
            g3_ready = split_prod && l > 1;
            if (!g16) CK("ln_bwd g", g, (size_t)M * D * 4);
            if (g_at) CK("ln_bwd g_at", g_at, (size_t)M * D * atb);
            if (student) CK("ln_bwd dmask", T.dmask, (size_t)M * 4);
            prepped = true;
        }
    }
    if (S.pool.wpending) { DYT_HIP_CHECK(hipStreamWaitEvent(s, S.pool.ev_wj, 0)); S.pool.wpending = false; }
    if (c->learn_scale && fix_hi > 0) {
        hipLaunchKernelGGL(learn_scale_fixup_kernel, dim3(fix_hi), dim3(256), 0, s, S.gscr, trainable, grad, c->layer_stride, c->off_uw, c->off_ub, c->off_sc, r, 0);
        DYT_HIP_CHECK(hipGetLastError());
    }
    return DYT_OK;
}

extern "C" int dyt_backward(dyt_ctx* c, int slot, const float* dlogits, const float* dtoken_select, const float* dtok,
                            const float* dtoken_logits, float* grad_flat, void* stream) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    if (slot < 0 || slot >= c->cfg.slots) { set_error("slot %d out of range", slot); return DYT_ERR_ARG; }
    return backward_impl(c, slot, c->slots[slot].trainable, dlogits, dtoken_select, dtok, dtoken_logits, grad_flat,
                         static_cast<hipStream_t>(stream));
}

extern "C" int dyt_loss(dyt_ctx* c, int slot_student, const float* logits_s, const float* logits_t, const int64_t* targets,
                        int batch, float token_target_ratio, float token_loss_ratio, float token_minimal,
                        float token_minimal_weight, float* dlogits_s, float* dlogits_t, float* out_losses, float* dtok,
                        void* stream) {
    if (!c || !logits_s || !logits_t || !targets || !dlogits_s || !dlogits_t || !out_losses || !dtok) {
        set_error("null argument");
        return DYT_ERR_ARG;
    }
    if (slot_student < 0 || slot_student >= c->cfg.slots) { set_error("slot out of range"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    LossArgs a;
    a.logits_s = logits_s; a.logits_t = logits_t; a.targets = targets; a.counts = c->slots[slot_student].counts;
    a.batch = batch; a.C = c->cfg.num_classes; a.depth = c->cfg.depth;
    a.count_batch = batch * c->frames;   // video: `batch` clips, the gates were evaluated on batch * t frames
    a.target_ratio = token_target_ratio; a.loss_ratio = token_loss_ratio; a.token_minimal = token_minimal;
    a.token_minimal_weight = token_minimal_weight;
    a.dlogits_s = dlogits_s; a.dlogits_t = dlogits_t; a.out_losses = out_losses; a.dtok = dtok;
    a.scratch = c->loss_part;
    if (batch * c->frames > c->cfg.max_batch) { set_error("batch %d exceeds max_batch", batch); return DYT_ERR_ARG; }
    if (c->soft_targets) {
        if (c->soft_batch != batch) { set_error("soft targets were set for %d rows, this loss has %d", c->soft_batch, batch); return DYT_ERR_STATE; }
        a.soft = c->soft_targets;
    }
    return launch_loss(a, s);
}

extern "C" int dyt_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, int step,
                         float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                         void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || numel < 1 || step < 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    return launch_adamw(param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale,
                        static_cast<hipStream_t>(stream));
}

// The same update, guarded like the reference's GradScaler.step (misc.py:256-272): if grad holds an inf / NaN (a 16-bit operand
// overflowed somewhere in the step) parameters and moments are left untouched and the skip is counted.  Nothing returns to the host:
// state (device int32[4], zero-initialised by the caller, owned by the optimizer) = {updates applied, updates skipped, flag of this
// call, reserved}; the bias corrections use state[0] + 1 as the step.
extern "C" int dyt_adamw_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, int32_t* state,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || numel < 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    return launch_adamw_guarded(param, grad, exp_avg, exp_avg_sq, numel, state, lr, beta1, beta2, eps, weight_decay, grad_scale,
                                static_cast<hipStream_t>(stream));
}

extern "C" int dyt_step_fwd_bwd(dyt_ctx* c, const float* images, const int64_t* targets, int batch, int flags,
                                const float* trainable, const float* g1, const float* g2, const uint8_t* keep_mask,
                                uint64_t seed, float token_target_ratio, float token_loss_ratio, float token_minimal,
                                float token_minimal_weight, float* grad_flat, float* out_losses, float* logits_s,
                                float* logits_t, float* token_select, void* stream) {
    if (!c || !targets || !grad_flat || !out_losses) { set_error("null argument"); return DYT_ERR_ARG; }
    if (c->cfg.slots < 2) { set_error("dyt_step_fwd_bwd needs 2 slots"); return DYT_ERR_STATE; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int depth = c->cfg.depth;
    const size_t nz = (size_t)depth * batch * NP;        // per-pass noise stride
    const size_t kz = (size_t)depth * batch * NT * c->cfg.ffn_num;
    float* ls = logits_s ? logits_s : c->logits_s;
    float* lt = logits_t ? logits_t : c->logits_t;
    const int fl = (flags & (DYT_F_MASKED_DENSE | DYT_F_DEVICE_SEED)) | DYT_F_TRAINING | DYT_F_SAVE;
    // Two-stream schedule: student pass on the caller's stream, teacher pass on a side stream
    // (fork/join with events; graph-capturable).  Profiling mode runs serially for clean per-kernel times.
    const bool par = c->overlap && c->ov_pass && !c->prof;
    if (par && !c->side) {
        // measurement hook (tools/probes/determinism_cumask.py): DYT_DBG_SIDE_CU_MASK = "cu" | "xcd" pins the teacher pass to
        // the odd CU octets / the upper four XCDs (mask bit i -> XCD i % 8), the probe pins the caller's stream to the rest
        const char* dbg_mask = getenv("DYT_DBG_SIDE_CU_MASK");
        if (dbg_mask && (dbg_mask[0] == 'c' || dbg_mask[0] == 'x' || dbg_mask[0] == 'i' || dbg_mask[0] == 'a')) {
            uint32_t words[8];
            for (int i = 0; i < 8; ++i) words[i] = dbg_mask[0] == 'c' ? 0xFF00FF00u : (dbg_mask[0] == 'x' ? 0xF0F0F0F0u : (dbg_mask[0] == 'a' ? 0xF8F8F8F8u : 0x00FFFFFFu));   // a: five XCDs for the (heavier) teacher pass
            DYT_HIP_CHECK(hipExtStreamCreateWithCUMask(&c->side, 8, words));
        } else
        {   // measurement knob DYT_SIDE_PRIORITY: the teacher pass's stream at another priority (-1 = higher than the caller's stream, 1 = lower)
            const char* pr = getenv("DYT_SIDE_PRIORITY");
            if (pr) {
                int lo = 0, hi = 0;
                DYT_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least urgent (numerically greatest), hi = most urgent
                const int want = atoi(pr) < 0 ? hi : (atoi(pr) > 0 ? lo : 0);
                DYT_HIP_CHECK(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, want));
            } else
        DYT_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        }
        DYT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        DYT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    hipStream_t s2 = par ? c->side : s;
    if (batch % c->frames != 0) { set_error("video model: batch %d is not a multiple of frames %d", batch, c->frames); return DYT_ERR_ARG; }
    int rc = dbg_ck_reset(s);
    if (rc) return rc;
    rc = prep_adapters(c, trainable, s);
    if (rc) return rc;
    rc = prep_pool(c, trainable, s);
    if (rc) return rc;
    if (par) { DYT_HIP_CHECK(hipEventRecord(c->ev_fork, s)); DYT_HIP_CHECK(hipStreamWaitEvent(s2, c->ev_fork, 0)); }
    {
        // hipGraph stream capture (ROCm 7.2): a stream that forks from a stream which is itself a fork of the capture's
        // origin stream crashes hipStreamEndCapture (bisected on MI355X: origin -> side is fine, origin -> branch is fine,
        // side -> branch is not).  While capturing, the teacher pass (on the side stream) therefore keeps its adapter
        // branch on its own stream; the graph still carries the two passes and the student's branch as parallel chains.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        DYT_HIP_CHECK(hipStreamIsCapturing(s, &cs));
        c->slots[1].no_branch = par && cs == hipStreamCaptureStatusActive;
        c->slots[0].no_branch = false;
    }
    // The two passes see the same images and the same frozen weights, and nothing trainable or random sits
    // in front of block 0's attention branch: the teacher pass reuses the student's embedding, LN1, qkv,
    // attention and proj of block 0 (its block-0 `u` pointers alias the student's for this step).
    const bool share = c->share_block0;
    if (share && !c->ev_b0) DYT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_b0, hipEventDisableTiming));
    {
        Slot& S0 = c->slots[0]; Slot& S1 = c->slots[1];
        S0.L[0].u = S0.u0_own; S0.L[0].u_at = S0.u0_at_own;
        S1.L[0].u = share ? S0.u0_own : S1.u0_own;
        S1.L[0].u_at = share ? S0.u0_at_own : S1.u0_at_own;
        S0.L[0].u16 = S0.u0_16_own; S1.L[0].u16 = share ? S0.u0_16_own : S1.u0_16_own;
        S0.L[0].ln_part = S0.part0_own; S1.L[0].ln_part = share ? S0.part0_own : S1.part0_own;
    }
    rc = forward_impl(c, 0, images, batch, fl, trainable, g1, g2, keep_mask, seed, ls, token_select, nullptr, false, s,
                      nullptr, share ? c->ev_b0 : nullptr, nullptr);
    if (rc) return rc;
    // the teacher pass draws its own noise in the reference (mask discarded): only the dropout stream matters
    rc = forward_impl(c, 1, images, batch, fl | DYT_F_COMPLETE, trainable, g1 ? g1 + nz : nullptr, g2 ? g2 + nz : nullptr,
                      keep_mask ? keep_mask + kz : nullptr, seed, lt, nullptr, nullptr, false, s2,
                      share ? &c->slots[0] : nullptr, nullptr, (share && par) ? c->ev_b0 : nullptr);
    if (rc) return rc;
    if (par) { DYT_HIP_CHECK(hipEventRecord(c->ev_join, s2)); DYT_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0)); }
    rc = dyt_loss(c, 0, ls, lt, targets, batch / c->frames, token_target_ratio, token_loss_ratio, token_minimal, token_minimal_weight,
                  c->dl_s, c->dl_t, out_losses, c->dtok, stream);
    if (rc) return rc;
    float* gt = par ? c->grad2 : grad_flat;  // teacher-pass gradients
    // measurement hook: DYT_DBG_TEACHER_NAN=1 feeds the teacher's backward pass NaN (every value it computes or stores is NaN) and
    // leaves the two passes' gradients unsummed: any NaN in grad_flat (the student's) is a write across the passes
    static const bool dbg_tnan = getenv("DYT_DBG_TEACHER_NAN") && atoi(getenv("DYT_DBG_TEACHER_NAN"));
    if (dbg_tnan && par) DYT_HIP_CHECK(hipMemsetAsync(c->dl_t, 0xFF, (size_t)batch * c->cfg.num_classes * sizeof(float), s));
    if (par) { DYT_HIP_CHECK(hipEventRecord(c->ev_fork, s)); DYT_HIP_CHECK(hipStreamWaitEvent(s2, c->ev_fork, 0)); }
    if (!(flags & DYT_F_ACCUM_GRAD)) DYT_HIP_CHECK(hipMemsetAsync(grad_flat, 0, (size_t)c->n_train * sizeof(float), s));
    if (par) DYT_HIP_CHECK(hipMemsetAsync(c->grad2, 0, (size_t)c->n_train * sizeof(float), s2));
    // Chunked all-reduce support (DDP fires its buckets inside loss.backward(), misc.py:258-259): the backward runs
    // block 11 -> 0, so the gradients of the head and of blocks >= depth/2 -- a contiguous tail of the flat buffer --
    // are final half-way through.  ev_upper marks that point (both passes summed) for dyt_stream_wait_grads().
    const int split = depth / 2;
    const int64_t up_off = (int64_t)split * c->layer_stride, up_n = c->n_train - up_off;
    if (!c->ev_upper) {
        DYT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_half_s, hipEventDisableTiming));
        DYT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_half_t, hipEventDisableTiming));
        DYT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_upper, hipEventDisableTiming));
        DYT_HIP_CHECK(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    }
    g_dbg_in_backward = 1;
    rc = backward_impl(c, 0, trainable, c->dl_s, nullptr, c->dtok, nullptr, grad_flat, s, par ? c->ev_half_s : nullptr, split);
    if (rc) { g_dbg_in_backward = 0; return rc; }
    if (par && c->ov_bwd_serial) { DYT_HIP_CHECK(hipEventRecord(c->ev_fork, s)); DYT_HIP_CHECK(hipStreamWaitEvent(s2, c->ev_fork, 0)); }
    rc = backward_impl(c, 1, trainable, c->dl_t, nullptr, nullptr, nullptr, gt, s2, par ? c->ev_half_t : c->ev_upper, split);
    g_dbg_in_backward = 0;
    if (rc) return rc;
    if (par) {
        // upper part: summed on the aux stream as soon as both passes have left block `split`
        DYT_HIP_CHECK(hipStreamWaitEvent(c->aux, c->ev_half_s, 0));
        DYT_HIP_CHECK(hipStreamWaitEvent(c->aux, c->ev_half_t, 0));
        if (!dbg_tnan) rc = launch_reduce_partials(c->grad2 + up_off, 1, 0, grad_flat + up_off, (int)up_n, 1.0f, c->aux);
        if (rc) return rc;
        DYT_HIP_CHECK(hipEventRecord(c->ev_upper, c->aux));
        // lower part: after the teacher pass has finished
        DYT_HIP_CHECK(hipEventRecord(c->ev_join, s2));
        DYT_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
        if (!dbg_tnan) rc = launch_reduce_partials(c->grad2, 1, 0, grad_flat, (int)up_off, 1.0f, s);  // grad_flat[lower] += grad2[lower]
        if (rc) return rc;
        DYT_HIP_CHECK(hipStreamWaitEvent(s, c->ev_upper, 0));   // the caller's stream owns the whole buffer on return
    }
    c->upper_recorded = true;
    if (flags & DYT_F_DEVICE_SEED) rc = launch_seed_advance(c->seed_dev, s);
    return rc;
}

extern "C" int dyt_seed(dyt_ctx* c, uint64_t seed, void* stream) {
    if (!c) { set_error("null ctx"); return DYT_ERR_ARG; }
    return launch_seed_set(c->seed_dev, seed, static_cast<hipStream_t>(stream));
}

extern "C" int dyt_grad_part(const dyt_ctx* c, int part, int64_t* offset, int64_t* numel) {
    if (!c || !offset || !numel || part < 0 || part > 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    const int64_t up_off = (int64_t)(c->cfg.depth / 2) * c->layer_stride;
    if (part == 0) { *offset = up_off; *numel = c->n_train - up_off; }
    else { *offset = 0; *numel = up_off; }
    return DYT_OK;
}

extern "C" int dyt_stream_wait_grads(dyt_ctx* c, int part, void* stream) {
    if (!c || part != 0) { set_error("dyt_stream_wait_grads: part 0 (head + upper blocks) is the only early part"); return DYT_ERR_ARG; }
    if (!c->upper_recorded) { set_error("no dyt_step_fwd_bwd has been enqueued yet"); return DYT_ERR_STATE; }
    DYT_HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->ev_upper, 0));
    return DYT_OK;
}

// ------------------------------------------------------------------------------------------
// gradient all-reduce on RCCL, behind the ABI (reference: DistributedDataParallel's bucket all-reduce inside loss.backward(),
// main_image.py:280-282 / misc.py:258-259).  librccl is NOT a link-time dependency: the symbol binds at load time to the RCCL that
// the host process already has (PyTorch's, promoted to the global scope by _lib.py, or the binder's own); absent -> an error.
// ------------------------------------------------------------------------------------------
extern "C" int ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm,
                             hipStream_t stream) __attribute__((weak));
extern "C" const char* ncclGetErrorString(int result) __attribute__((weak));

extern "C" int dyt_allreduce_grads(dyt_ctx* c, void* rccl_comm, float* grad_flat, void* comm_stream, void* stream) {
    if (!c || !rccl_comm || !grad_flat) { set_error("null argument"); return DYT_ERR_ARG; }
    if (!ncclAllReduce) { set_error("RCCL is not loaded in this process (ncclAllReduce unresolved)"); return DYT_ERR_STATE; }
    constexpr int kNcclFloat32 = 7, kNcclSum = 0;
    hipStream_t s = static_cast<hipStream_t>(stream), cs = static_cast<hipStream_t>(comm_stream);
    const int64_t up_off = (int64_t)(c->cfg.depth / 2) * c->layer_stride, up_n = c->n_train - up_off;
    auto chk = [](int rc) {
        if (rc != 0) { set_error("ncclAllReduce failed: %s", ncclGetErrorString ? ncclGetErrorString(rc) : "?"); return DYT_ERR_HIP; }
        return 0;
    };
    if (cs && cs != s && c->upper_recorded) {
        // part 0 (head + upper blocks): final half-way through the backward pass -> reduced on the communication stream while the
        // frozen-backbone backward of the lower blocks is still running on `stream`; part 1 follows on `stream`
        if (!c->ev_comm) DYT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
        DYT_HIP_CHECK(hipStreamWaitEvent(cs, c->ev_upper, 0));
        int rc = chk(ncclAllReduce(grad_flat + up_off, grad_flat + up_off, (size_t)up_n, kNcclFloat32, kNcclSum, rccl_comm, cs));
        if (rc) return rc;
        DYT_HIP_CHECK(hipEventRecord(c->ev_comm, cs));
        rc = chk(ncclAllReduce(grad_flat, grad_flat, (size_t)up_off, kNcclFloat32, kNcclSum, rccl_comm, s));
        if (rc) return rc;
        DYT_HIP_CHECK(hipStreamWaitEvent(s, c->ev_comm, 0));
        return DYT_OK;
    }
    return chk(ncclAllReduce(grad_flat, grad_flat, (size_t)c->n_train, kNcclFloat32, kNcclSum, rccl_comm, s));
}

extern "C" int dyt_clip_grad_norm(dyt_ctx* c, float* grad, int64_t numel, float max_norm, float pre_scale, float* norm_out,
                                  void* stream) {
    if (!c || !grad || numel < 1 || !(max_norm > 0.f)) { set_error("bad argument"); return DYT_ERR_ARG; }
    return launch_clip_grad_norm(grad, numel, max_norm, pre_scale, c->clip_scratch, norm_out, static_cast<hipStream_t>(stream));
}

extern "C" int dyt_debug_dispatch(dyt_ctx* c, int slot, int layer, int32_t* row_src, int32_t* dst_of, int32_t* counts,
                                  int32_t* total, void* stream) {
    if (!c || slot < 0 || slot >= c->cfg.slots || layer < 0 || layer >= c->cfg.depth) { set_error("bad slot / layer"); return DYT_ERR_ARG; }
    const Slot& S = c->slots[slot];
    if (S.batch < 1) { set_error("slot %d holds no pass", slot); return DYT_ERR_STATE; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t M = (size_t)S.batch * NT;
    const LayerS& L = S.L[layer];
    if (row_src) DYT_HIP_CHECK(hipMemcpyAsync(row_src, L.row_src, M * 4, hipMemcpyDeviceToDevice, s));
    if (dst_of) DYT_HIP_CHECK(hipMemcpyAsync(dst_of, L.dst_of, M * 4, hipMemcpyDeviceToDevice, s));
    if (counts) DYT_HIP_CHECK(hipMemcpyAsync(counts, S.counts + (size_t)layer * S.batch, (size_t)S.batch * 4, hipMemcpyDeviceToDevice, s));
    if (total) DYT_HIP_CHECK(hipMemcpyAsync(total, L.total, 4, hipMemcpyDeviceToDevice, s));
    return DYT_OK;
}

// Saved adapter bottleneck relu(down(u)) (times the dropout scale) of one block of a saved pass, as fp32 [rows, 64]; *rows_out = B*197,
// or B for the last block in the cls-only tail form.  Test accessor (tests/parity_rules.py checks which side of the ReLU a unit is on).
extern "C" int dyt_debug_dact(dyt_ctx* c, int slot, int layer, float* out, int* rows_out, void* stream) {
    if (!c || !out || slot < 0 || slot >= c->cfg.slots || layer < 0 || layer >= c->cfg.depth) { set_error("bad slot / layer"); return DYT_ERR_ARG; }
    const Slot& S = c->slots[slot];
    if (S.batch < 1) { set_error("slot %d holds no pass", slot); return DYT_ERR_STATE; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool tail = c->cls_tail && layer == c->cfg.depth - 1;
    const size_t rows = tail ? (size_t)S.batch : (size_t)S.batch * NT, n = rows * RP;
    const LayerS& L = S.L[layer];
    if (rows_out) *rows_out = (int)rows;
    if (S.saved16 && L.dact16) hipLaunchKernelGGL(to_f32_kernel<bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)L.dact16, out, n);
    else if (c->prec == DYT_PREC_FP32) DYT_HIP_CHECK(hipMemcpyAsync(out, L.d_act, n * 4, hipMemcpyDeviceToDevice, s));
    else hipLaunchKernelGGL(to_f32_kernel<bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)L.d_act, out, n);
    DYT_HIP_CHECK(hipGetLastError());
    return DYT_OK;
}

extern "C" int dyt_debug_drop_path(dyt_ctx* c, int slot, float* out, void* stream) {
    if (!c || !out || slot < 0 || slot >= c->cfg.slots) { set_error("bad slot"); return DYT_ERR_ARG; }
    const Slot& S = c->slots[slot];
    if (S.batch < 1 || !S.dp) { set_error("slot %d: the last pass ran without stochastic depth", slot); return DYT_ERR_STATE; }
    DYT_HIP_CHECK(hipMemcpyAsync(out, S.dp, (size_t)2 * c->cfg.depth * S.batch * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return DYT_OK;
}

// ------------------------------------------------------------------------------------------
// single-kernel entry points (unit tests).  These allocate scratch and synchronise: test-only.
// ------------------------------------------------------------------------------------------
extern "C" int dyt_gemm_bf16_raw(const void* a, const void* w, void* cmat, int M, int N, int K, int variant, void* stream) {
    if (!a || !w || !cmat) { set_error("null argument"); return DYT_ERR_ARG; }
    return launch_gemm_raw(a, w, cmat, M, N, K, variant, static_cast<hipStream_t>(stream));
}

// adapter weight-gradient kernel alone (unit tests / probes): out_w[c*r + j] += sum_m X[m][c] Y[m][j], out_xsum[c] += sum_m X[m][c],
// out_ysum[j] += sum_m Y[m][j]; X [M,768], Y [M,64] in the precision's operand type; partial = scratch of dyt_wgrad_scratch_floats(M)
extern "C" int64_t dyt_wgrad_scratch_floats(int M) { return (int64_t)((M + 511) / 512) * (D + 8) * 80; }
extern "C" int dyt_wgrad_raw(const void* X, const void* Y, int M, int r, int precision, float* partial, float* out_w, float* out_xsum,
                             float* out_ysum, void* stream) {
    if (!X || !Y || !partial || !out_w || M < 1 || r < 1 || r > RP) { set_error("bad argument"); return DYT_ERR_ARG; }
    WgradArgs a; a.X = X; a.Y = Y; a.M = M; a.r = r; a.partial = partial;
    a.out_w = out_w; a.sc = r; a.sj = 1; a.alpha = 1.0f; a.out_xsum = out_xsum; a.alpha_x = 1.0f; a.out_ysum = out_ysum; a.alpha_y = 1.0f;
    return launch_wgrad(precision, a, static_cast<hipStream_t>(stream));
}

extern "C" int dyt_gemm_f32_raw(const float* a, const float* w, float* cmat, int M, int N, int K, int variant, void* stream) {
    if (!a || !w || !cmat) { set_error("null argument"); return DYT_ERR_ARG; }
    return launch_gemm_f32_raw(a, w, cmat, M, N, K, variant, static_cast<hipStream_t>(stream));
}

extern "C" int dyt_debug_counters(uint64_t* out4, int reset) {
    if (!out4) { set_error("null argument"); return DYT_ERR_ARG; }
    return gemm_debug_counters(reinterpret_cast<unsigned long long*>(out4), reset);
}

extern "C" int dyt_layernorm(const float* x, const float* w, const float* b, float* out, int rows, void* stream) {
    if (!x || !w || !b || !out || rows < 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    return launch_ln_fwd_f32out(x, w, b, out, rows, static_cast<hipStream_t>(stream));
}

struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) hipFree(p); }
    void* get(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
};

extern "C" int dyt_linear(const float* a, const float* w, const float* bias, float* cmat, int M, int N, int K, int precision,
                          void* stream) {
    if (!a || !w || !cmat || M < 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    GemmArgs g; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out_f32 = cmat;
    Scratch sc;
    if (precision == 0) { g.A = a; g.W = w; }
    else {
        void* a2 = sc.get((size_t)M * K * 2); void* w2 = sc.get((size_t)N * K * 2);
        if (!a2 || !w2) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
        int rc = launch_convert(1, a, a2, (int64_t)M * K, s); if (rc) return rc;
        rc = launch_convert(1, w, w2, (int64_t)N * K, s); if (rc) return rc;
        g.A = a2; g.W = w2;
    }
    int rc = launch_gemm(precision, EPI_BIAS_F32, g, s);
    if (rc) return rc;
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}

// One nn.Linear through the split forms of the fp32 mode (unit tests / probes): form 3 = three IEEE-half products, 8 = hi * hi in f16 +
// fp8 correction products.  a [M,K], w [N,K], bias [N] or NULL, cmat [M,N] fp32.  Allocates scratch and synchronises: test-only.
extern "C" int dyt_linear_split(const float* a, const float* w, const float* bias, float* cmat, int M, int N, int K, int form, void* stream) {
    if (!a || !w || !cmat || M < 1 || (form != 3 && form != 8)) { set_error("bad argument"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    void* a3 = sc.get((size_t)((M + 255) / 256 * 256) * SPLIT_A * K * 2); void* w3 = sc.get((size_t)N * SPLIT_A * K * 2);
    int* ew = (int*)sc.get(16); unsigned* scr = (unsigned*)sc.get(16);
    if (!a3 || !w3 || !ew || !scr) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
    int rc = form == 8 ? launch_split_w_f8(w, w3, N, K, ew, scr, s) : launch_split3_w(w, w3, N, K, s);
    if (rc) return rc;
    GemmArgs g; g.A = a; g.W = w; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out_f32 = cmat; g.W3 = w3; g.a3 = a3;
    if (form == 8) { g.f8 = true; g.w_exp = ew; }
    rc = launch_gemm(0, EPI_BIAS_F32, g, s);
    if (rc) return rc;
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}

template <class AT>
static int attention_test(const float* qkv, float* out, const float* dout, float* dqkv, int B, int P, hipStream_t s) {
    Scratch sc;
    const size_t M = (size_t)B * NT;
    AT* q = (AT*)sc.get(M * D * sizeof(AT)); AT* k = (AT*)sc.get(M * D * sizeof(AT)); AT* v = (AT*)sc.get(M * D * sizeof(AT));
    AT* o = (AT*)sc.get(M * D * sizeof(AT));
    float* lse = (float*)sc.get((size_t)B * NH * NT * 4);
    if (!q || !k || !v || !o || !lse) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
    const size_t n3 = M * 3 * D;
    hipLaunchKernelGGL(qkv_split_kernel<AT>, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, qkv, q, k, v, B);
    int rc = launch_attn_fwd(P, q, k, v, o, lse, B, s);
    if (rc) return rc;
    hipLaunchKernelGGL(to_f32_kernel<AT>, dim3((unsigned)((M * D + 255) / 256)), dim3(256), 0, s, (const AT*)o, out, M * D);
    if (dout && dqkv) {
        AT* d_o = (AT*)sc.get(M * D * sizeof(AT)); AT* dq = (AT*)sc.get(n3 * sizeof(AT));
        float* delta = (float*)sc.get((size_t)B * NH * NT * 4);
        if (!d_o || !dq || !delta) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
        rc = launch_convert(P, dout, d_o, (int64_t)(M * D), s); if (rc) return rc;
        rc = launch_attn_bwd(P, q, k, v, o, d_o, lse, delta, dq, B, s); if (rc) return rc;
        hipLaunchKernelGGL(to_f32_kernel<AT>, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, (const AT*)dq, dqkv, n3);
    }
    DYT_HIP_CHECK(hipGetLastError());
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}

extern "C" int dyt_attention(const float* qkv, float* out, const float* dout, float* dqkv, int batch, int precision,
                             void* stream) {
    if (!qkv || !out || batch < 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    return precision == 0 ? attention_test<float>(qkv, out, dout, dqkv, batch, 0, s)
                          : attention_test<bf16>(qkv, out, dout, dqkv, batch, 1, s);
}

// ---- sub-module unit entries: the adapter and the gathered MLP alone, through the product's own kernels ----
namespace dyt {
// per image: kept-token list (ascending) and count from a {0,1} mask -- what gate_select_kernel leaves behind
__global__ __launch_bounds__(256) void mask_to_keep_kernel(const float* __restrict__ maskf, int* __restrict__ keep_local,
                                                           int* __restrict__ counts) {
    __shared__ int wave_cnt[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool keep = tid < NT && maskf[(size_t)b * NT + tid] != 0.f;
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (keep) keep_local[(size_t)b * NT + off + __popcll(bal & ((1ull << lane) - 1ull))] = tid;
    if (tid == 0) counts[b] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}
}  // namespace dyt

struct AdapterOps {   // AT copies of one adapter's weights in the layouts the GEMMs want (see prep_adapters_kernel)
    void *x_at, *down_w, *down_wT, *up_w, *up_wT, *d_act;
    float* down_b;
};
static int adapter_prepare(Scratch& sc, int P, const float* x, const float* down_w, const float* down_b, const float* up_w, int M, int r,
                           AdapterOps* o, hipStream_t s) {
    const size_t at = at_size(P);
    o->x_at = P == 0 ? (void*)x : sc.get((size_t)M * D * at);
    o->down_w = sc.get((size_t)RP * D * at); o->down_wT = sc.get((size_t)RP * D * at);
    o->up_w = sc.get((size_t)RP * D * at); o->up_wT = sc.get((size_t)RP * D * at);
    o->d_act = sc.get((size_t)M * RP * at);
    o->down_b = (float*)sc.get(RP * sizeof(float));
    if (!o->x_at || !o->down_w || !o->down_wT || !o->up_w || !o->up_wT || !o->d_act || !o->down_b) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
    int rc = 0;
    if (P != 0) rc = launch_convert(P, x, o->x_at, (int64_t)M * D, s);
    if (!rc) rc = launch_pad_convert(P, down_w, o->down_w, r, D, RP, D, s);              // [RP,768], rows >= r zero
    if (!rc) rc = launch_transpose_convert(P, down_w, o->down_wT, r, D, D, RP, s);       // [768,RP]
    if (!rc) rc = launch_pad_convert(P, up_w, o->up_w, D, r, D, RP, s);                  // [768,RP], cols >= r zero
    if (!rc) rc = launch_transpose_convert(P, up_w, o->up_wT, D, r, RP, D, s);           // [RP,768]
    if (rc) return rc;
    DYT_HIP_CHECK(hipMemsetAsync(o->down_b, 0, RP * sizeof(float), s));
    DYT_HIP_CHECK(hipMemcpyAsync(o->down_b, down_b, (size_t)r * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
static int adapter_down(int P, const AdapterOps& o, int M, int r, float drop_p, const uint8_t* keep, uint64_t seed, hipStream_t s) {
    GemmArgs a; a.A = o.x_at; a.W = o.down_w; a.M = M; a.N = RP; a.K = D; a.bias = o.down_b; a.out_at = o.d_act; a.r = r;
    a.drop_p = drop_p; a.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f; a.keep = keep; a.seed = seed; a.subseq = 1;
    return launch_gemm(P, EPI_AD_DOWN, a, s);
}

// Adapter.forward (reference models/dynamic_adapter.py:120-140): out = [residual +] scale * up(dropout_p(relu(down(x))))
extern "C" int dyt_adapter_fwd(const float* x, const float* down_w, const float* down_b, const float* up_w, const float* up_b,
                               const float* residual, float* out, int M, int r, float scale, float drop_p, const uint8_t* keep_mask,
                               uint64_t seed, int precision, void* stream) {
    if (!x || !down_w || !down_b || !up_w || !up_b || !out || M < 1 || r < 1 || r > RP || (precision != 0 && precision != 1)) {
        set_error("bad argument");
        return DYT_ERR_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    AdapterOps o;
    int rc = adapter_prepare(sc, precision, x, down_w, down_b, up_w, M, r, &o, s);
    if (rc) return rc;
    float* zero = nullptr;
    if (!residual) {
        zero = (float*)sc.get((size_t)M * D * sizeof(float));
        if (!zero) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
        DYT_HIP_CHECK(hipMemsetAsync(zero, 0, (size_t)M * D * sizeof(float), s));
    }
    rc = adapter_down(precision, o, M, r, drop_p, keep_mask, seed, s);
    if (rc) return rc;
    GemmArgs a; a.A = o.d_act; a.W = o.up_w; a.M = M; a.N = D; a.K = RP; a.bias = up_b; a.resid = residual ? residual : zero;
    a.out_f32 = out; a.scale = scale;
    rc = launch_gemm(precision, EPI_AD_UP, a, s);
    if (rc) return rc;
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}

// Its backward for an upstream gradient dout [M,768] (the same draws): dx [M,768] (may be NULL) and the four parameter gradients,
// ACCUMULATED into d_down_w [r,768], d_down_b [r], d_up_w [768,r], d_up_b [768].
extern "C" int dyt_adapter_bwd(const float* x, const float* down_w, const float* down_b, const float* up_w, const float* dout, float* dx,
                               float* d_down_w, float* d_down_b, float* d_up_w, float* d_up_b, int M, int r, float scale, float drop_p,
                               const uint8_t* keep_mask, uint64_t seed, int precision, void* stream) {
    if (!x || !down_w || !down_b || !up_w || !dout || !d_down_w || !d_down_b || !d_up_w || !d_up_b || M < 1 || r < 1 || r > RP ||
        (precision != 0 && precision != 1)) {
        set_error("bad argument");
        return DYT_ERR_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int P = precision;
    const size_t at = at_size(P);
    Scratch sc;
    AdapterOps o;
    int rc = adapter_prepare(sc, P, x, down_w, down_b, up_w, M, r, &o, s);
    if (rc) return rc;
    rc = adapter_down(P, o, M, r, drop_p, keep_mask, seed, s);   // recompute the bottleneck activations (relu / dropout pattern)
    if (rc) return rc;
    void* g_at = P == 0 ? (void*)dout : sc.get((size_t)M * D * at);
    void* ddz = sc.get((size_t)M * RP * at);
    float* p1 = (float*)sc.get((size_t)dyt_wgrad_scratch_floats(M) * sizeof(float));
    float* p2 = (float*)sc.get((size_t)dyt_wgrad_scratch_floats(M) * sizeof(float));
    if (!g_at || !ddz || !p1 || !p2) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
    if (P != 0) { rc = launch_convert(P, dout, g_at, (int64_t)M * D, s); if (rc) return rc; }
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    {
        GemmArgs a; a.A = g_at; a.W = o.up_wT; a.M = M; a.N = RP; a.K = D; a.aux_at = o.d_act; a.out_at = ddz; a.scale = scale; a.inv_keep = inv_keep;
        rc = launch_gemm(P, EPI_AD_DGRAD_UP, a, s); if (rc) return rc;
    }
    WgradArgs w[2];
    w[0].X = g_at; w[0].Y = o.d_act; w[0].M = M; w[0].r = r; w[0].partial = p1; w[0].out_w = d_up_w; w[0].sc = r; w[0].sj = 1;
    w[0].alpha = scale; w[0].out_xsum = d_up_b; w[0].alpha_x = scale;
    w[1].X = o.x_at; w[1].Y = ddz; w[1].M = M; w[1].r = r; w[1].partial = p2; w[1].out_w = d_down_w; w[1].sc = 1; w[1].sj = D;
    w[1].alpha = 1.0f; w[1].out_xsum = nullptr; w[1].alpha_x = 0.f; w[1].out_ysum = d_down_b; w[1].alpha_y = 1.0f;
    rc = launch_wgrad(P, w, 2, s); if (rc) return rc;
    if (dx) {
        GemmArgs a; a.A = ddz; a.W = o.down_wT; a.M = M; a.N = D; a.K = RP; a.out_f32 = dx; a.accumulate = 0;
        rc = launch_gemm(P, EPI_STORE_F32, a, s); if (rc) return rc;
    }
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}

// The token-gathered MLP of block `layer` with the context's frozen weights (reference models/model_speed_test.py:297-305):
// x [B*197,768] += scatter(fc2(gelu(fc1(LN2(gather(u, mask)))))) for the tokens whose mask is non-zero; u, x fp32, mask [B*197].
extern "C" int dyt_mlp_gathered_fwd(dyt_ctx* c, int layer, const float* u, const float* mask, float* x, int batch, int32_t* total_out,
                                    void* stream) {
    if (!c || !u || !mask || !x || layer < 0 || layer >= c->cfg.depth || batch < 1 || batch > c->cfg.max_batch) { set_error("bad argument"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int P = c->prec, M = batch * NT;
    const LayerW& W = c->W[layer];
    Scratch sc;
    int* keep_local = (int*)sc.get((size_t)M * 4); int* counts = (int*)sc.get((size_t)batch * 4); int* total = (int*)sc.get(16);
    int* row_src = (int*)sc.get((size_t)M * 4); int* dst_of = (int*)sc.get((size_t)M * 4);
    float2* st = (float2*)sc.get((size_t)M * sizeof(float2));
    void* xn = sc.get((size_t)M * D * c->at); void* h1 = sc.get((size_t)M * DM * c->at);
    if (!keep_local || !counts || !total || !row_src || !dst_of || !st || !xn || !h1) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
    hipLaunchKernelGGL(mask_to_keep_kernel, dim3(batch), dim3(256), 0, s, mask, keep_local, counts);
    DYT_HIP_CHECK(hipGetLastError());
    int rc = launch_ln_gather(P, u, W.ln2_w, W.ln2_b, keep_local, counts, total, mask, xn, st, row_src, dst_of, batch, s);
    if (rc) return rc;
    {
        GemmArgs a; a.A = xn; a.W = W.fc1_w; a.Wp = W.fc1_wp; a.M = M; a.N = DM; a.K = D; a.m_dev = total; a.bias = W.fc1_b; a.out_at = h1;
        rc = launch_gemm(P, EPI_FC1, a, s); if (rc) return rc;
    }
    {
        GemmArgs a; a.A = h1; a.W = W.fc2_w; a.M = M; a.N = D; a.K = DM; a.m_dev = total; a.bias = W.fc2_b; a.out_f32 = x; a.row_map = row_src;
        rc = launch_gemm(P, EPI_FC2, a, s); if (rc) return rc;
    }
    if (total_out) DYT_HIP_CHECK(hipMemcpyAsync(total_out, total, 4, hipMemcpyDeviceToDevice, s));
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}

// Its backward for an upstream gradient dy [B*197,768] (the gradient w.r.t. x of dyt_mlp_gathered_fwd): du [B*197,768] += the gradient that
// reaches u THROUGH the gathered MLP (LN2 backward of fc1^T (gelu'(z) * (fc2^T dy)) for the kept tokens, nothing for the dropped ones;
// the residual path du += dy is the caller's).  Recomputes the forward up to gelu'(z); frozen weights: no weight gradients.
extern "C" int dyt_mlp_gathered_bwd(dyt_ctx* c, int layer, const float* u, const float* mask, const float* dy, float* du, int batch,
                                    void* stream) {
    if (!c || !u || !mask || !dy || !du || layer < 0 || layer >= c->cfg.depth || batch < 1 || batch > c->cfg.max_batch) { set_error("bad argument"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int P = c->prec, M = batch * NT;
    const float gs = P == 0 ? 1.0f : c->gs;
    const LayerW& W = c->W[layer];
    Scratch sc;
    int* keep_local = (int*)sc.get((size_t)M * 4); int* counts = (int*)sc.get((size_t)batch * 4); int* total = (int*)sc.get(16);
    int* row_src = (int*)sc.get((size_t)M * 4); int* dst_of = (int*)sc.get((size_t)M * 4);
    float2* st = (float2*)sc.get((size_t)M * sizeof(float2));
    void* xn = sc.get((size_t)M * D * c->at); void* h1 = sc.get((size_t)M * DM * c->at); void* gp = sc.get((size_t)M * DM * c->at);
    void* g_at = P == 0 ? nullptr : sc.get((size_t)M * D * c->at);
    void* dZ = sc.get((size_t)M * DM * c->at); void* dA2 = sc.get((size_t)M * D * c->at);
    float* partial = (float*)sc.get((size_t)((M + 31) / 32) * (D + 1) * sizeof(float));
    if (!keep_local || !counts || !total || !row_src || !dst_of || !st || !xn || !h1 || !gp || (P != 0 && !g_at) || !dZ || !dA2 || !partial) {
        set_error("scratch alloc failed");
        return DYT_ERR_HIP;
    }
    hipLaunchKernelGGL(mask_to_keep_kernel, dim3(batch), dim3(256), 0, s, mask, keep_local, counts);
    DYT_HIP_CHECK(hipGetLastError());
    int rc = launch_ln_gather(P, u, W.ln2_w, W.ln2_b, keep_local, counts, total, mask, xn, st, row_src, dst_of, batch, s);
    if (rc) return rc;
    {
        GemmArgs a; a.A = xn; a.W = W.fc1_w; a.Wp = W.fc1_wp; a.M = M; a.N = DM; a.K = D; a.m_dev = total; a.bias = W.fc1_b; a.out_at = h1; a.out_at2 = gp;
        rc = launch_gemm(P, EPI_FC1, a, s); if (rc) return rc;
    }
    if (P != 0) {
        BwdPrepArgs a; a.g = dy; a.h = nullptr; a.dst_of = nullptr; a.row_mask = nullptr; a.g_at = g_at; a.dH = nullptr; a.dmask = nullptr; a.M = M; a.gs = gs;
        rc = launch_bwd_prep(P, a, s); if (rc) return rc;
    }
    {
        GemmArgs a; a.A = P == 0 ? (const void*)dy : (const void*)g_at; a.W = W.fc2_wT; a.Wp = W.fc2_wTp; a.M = M; a.N = DM; a.K = D; a.m_dev = total;
        a.aux_at = gp; a.a_map = row_src; a.out_at = dZ;
        rc = launch_gemm(P, EPI_GELU_BWD, a, s); if (rc) return rc;
    }
    {
        GemmArgs a; a.A = dZ; a.W = W.fc1_wT; a.Wp = W.fc1_wTp; a.M = M; a.N = D; a.K = DM; a.m_dev = total; a.out_at = dA2;
        rc = launch_gemm(P, EPI_STORE_AT, a, s); if (rc) return rc;
    }
    {
        TokBwdArgs a;
        a.du = du; a.dA2 = dA2; a.dst_of = dst_of; a.u = u; a.stats2 = st; a.ln2_w = W.ln2_w; a.gate_w = nullptr; a.soft = nullptr; a.maskf = mask;
        a.dmask = nullptr; a.dtoken_select = nullptr; a.dtoken_logits = nullptr; a.dtok = nullptr; a.out_stride = 0; a.training = 0; a.tau = 1.0f;
        a.du_at = nullptr; a.partial = partial; a.M = M; a.write_du = 1; a.gs = gs; a.inv_gs = 1.0f / gs;
        int nblk = 0;
        rc = launch_tok_bwd(P, a, &nblk, s); if (rc) return rc;
    }
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}

extern "C" int dyt_gate_compact(const float* u, const float* w, const float* b, const float* g1, const float* g2, int batch,
                                int training, float tau, float threshold, float* mask, float* logits, int32_t* keep_idx,
                                int32_t* counts, int32_t* total, void* stream) {
    if (!u || !w || !b || !mask || !logits || !keep_idx || !counts || !total || batch < 1) { set_error("bad argument"); return DYT_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    const size_t M = (size_t)batch * NT;
    float* soft = (float*)sc.get(M * 4); float* maskf = (float*)sc.get(M * 4);
    int* keep_local = (int*)sc.get(M * 4); int* dst_of = (int*)sc.get(M * 4);
    float* xn = (float*)sc.get(M * D * 4); float2* st = (float2*)sc.get(M * sizeof(float2)); float* ones = (float*)sc.get(2 * D * 4);
    if (!soft || !maskf || !keep_local || !dst_of || !xn || !st || !ones) { set_error("scratch alloc failed"); return DYT_ERR_HIP; }
    GateArgs ga;
    ga.u = u; ga.w = w; ga.b = b; ga.g1 = g1; ga.g2 = g2; ga.batch = batch; ga.training = training; ga.tau = tau;
    ga.threshold = threshold; ga.seed = 0; ga.subseq = 0; ga.soft = soft; ga.maskf = maskf; ga.out_select = mask;
    ga.out_logits = logits; ga.out_stride = NP; ga.keep_local = keep_local; ga.counts = counts;
    int rc = launch_gate(ga, s); if (rc) return rc;
    // flat ascending list of kept rows (= nonzero() of model_speed_test.py:300) built on the DEVICE by the product's own
    // gather kernel: its row_src output is that list, its device-side total the length (the LayerNorm it also computes is discarded)
    DYT_HIP_CHECK(hipMemsetAsync(keep_idx, 0xff, M * 4, s));
    DYT_HIP_CHECK(hipMemsetAsync(ones, 0, 2 * D * 4, s));
    rc = launch_ln_gather(0, u, ones, ones + D, keep_local, counts, total, maskf, xn, st, keep_idx, dst_of, batch, s);
    if (rc) return rc;
    DYT_HIP_CHECK(hipStreamSynchronize(s));
    return DYT_OK;
}
