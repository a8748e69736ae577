// Row-wise and small kernels of the DyT hot path (gfx950): LayerNorm fwd/bwd, the token
// dispatcher (gate) with index compaction, gather, adapter/gate weight gradients, head, the
// step loss and AdamW.  All are HBM-bound: one 64-lane wave owns one 768-channel token row
// (3 x float4 per lane, fully coalesced), reductions are wave shuffles.
//
// Reference ops replaced (paths relative to the reference root):
//   nn.LayerNorm(eps=1e-6)                      models/vision_transformer_IN21K.py:110,123,314
//   TokenSelect.forward / _gumbel_sigmoid       models/dynamic_adapter.py:25-77
//   nonzero + gather of model_speed_test        models/model_speed_test.py:297-304
//   head / cls pooling                          models/vision_transformer_IN21K.py:375-380
//   CE + KL + AdaLoss                           engine_finetune.py:52-63, models/losses.py:48-84
//   torch.optim.AdamW                           main_image.py:285
#include <stdlib.h>
#include <algorithm>
#include "kernels.h"
#include "rowhelp.h"

namespace dyt {

#define LAUNCH_CHECK() DYT_HIP_CHECK(hipGetLastError())

// A/B switches of the probes (tools/probes/README.md): which of the full drains of ln_bwd / tok_bwd are compiled in
#if defined(DYT_NOPIN_LN) || defined(DYT_NOPIN_ROWS)
#define LN_ROWPIN(x)
#else
#define LN_ROWPIN(x) (x).landed()
#endif
#if defined(DYT_NOPIN_LN) || defined(DYT_NOPIN_SCALARS)
#define LN_SCALPIN3(a, b, c)
#else
#define LN_SCALPIN3(a, b, c) DYT_PIN3(a, b, c)
#endif
#if defined(DYT_NOPIN_TOK) || defined(DYT_NOPIN_ROWS)
#define TK_ROWPIN(x)
#else
#define TK_ROWPIN(x) (x).landed()
#endif
#if defined(DYT_NOPIN_TOK) || defined(DYT_NOPIN_SCALARS)
#define TK_SCALPIN3(a, b, c)
#define TK_SCALPIN4(a, b, c, d)
#else
#define TK_SCALPIN3(a, b, c) DYT_PIN3(a, b, c)
#define TK_SCALPIN4(a, b, c, d) DYT_PIN4(a, b, c, d)
#endif

// ------------------------------------------------------------------------------------------
// LayerNorm forward / backward
// ------------------------------------------------------------------------------------------
template <class AT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, AT* __restrict__ out,
                                                     float2* __restrict__ stats, int rows, bf16* __restrict__ out3, int f8) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    Row12 xr, wr, br;
    xr.load(x + (size_t)row * D, lane);
    wr.load(w, lane);
    br.load(b, lane);
    xr.landed(); wr.landed(); br.landed();   // one full drain for the whole load group (DYT_PIN*, dyt_common.h)
    const float2 st = ln_stats(xr);
#pragma unroll
    for (int i = 0; i < 12; ++i) xr.v[i] = (xr.v[i] - st.x) * st.y * wr.v[i] + br.v[i];
    if (out3) {   // fp32 split form: the row as the 16-bit hi / hi / lo operand of the next GEMM
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (f8) store4_split_f8(out3 + (size_t)row * SPLIT_A * D, D, i * 256 + lane * 4, xr.v[4 * i], xr.v[4 * i + 1], xr.v[4 * i + 2], xr.v[4 * i + 3]);
            else store4_split3(out3 + (size_t)row * SPLIT_A * D + i * 256 + lane * 4, D, xr.v[4 * i], xr.v[4 * i + 1], xr.v[4 * i + 2], xr.v[4 * i + 3]);
    } else {
        xr.store(out + (size_t)row * D, lane);
    }
    if (stats && lane == 0) stats[row] = st;
}

// dx = base + LNbwd(dy).  Optionally also does the NEXT (lower) block's backward prep on the row it just
// produced (saves re-reading the 77 MB gradient): AT copy of dx, and <dx, h_next> for the gate gradient.
template <class AT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const AT* __restrict__ dy, const float* __restrict__ x,
                                                     const float2* __restrict__ stats, const float* __restrict__ w,
                                                     const float* __restrict__ base, float* __restrict__ dx, int rows,
                                                     AT* __restrict__ g_at, const AT* __restrict__ h_next,
                                                     const int* __restrict__ dst_of_next, float* __restrict__ dmask_next,
                                                     float gs, float inv_gs, bf16* __restrict__ out3, float s3, int hi3,
                                                     const AT* __restrict__ base_at) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // every load of the row is issued up front and completed by ONE full drain that names all of them (DYT_PIN*, dyt_common.h)
    // base_at (round 6, fp16 mode): the gradient stream this row's LayerNorm gradient is added to arrives as the 16-bit, gs-scaled copy the
    // kernel before wrote for its GEMM (tok_bwd's du_at) instead of as an fp32 [M,768] stream; dx == null: no fp32 copy is written either
    Row12 g, xr, wr, br;
    g.load_at(dy + (size_t)row * D, lane);
    xr.load_nt(x + (size_t)row * D, lane);   // residual snapshot of the forward pass: last use
    wr.load(w, lane);
    float2 st = stats[row];
    int r = dst_of_next ? dst_of_next[row] : row;
    if (base_at) br.load_at(base_at + (size_t)row * D, lane);
    else if (base) br.load(base + (size_t)row * D, lane);
    LN_ROWPIN(g); LN_ROWPIN(xr); LN_ROWPIN(wr);
    LN_SCALPIN3(st.x, st.y, r);
    if (base || base_at) LN_ROWPIN(br);
    ln_bwd_row(g, xr, wr, st);
#pragma unroll
    for (int i = 0; i < 12; ++i) g.v[i] *= inv_gs;   // dy carried the 16-bit gradient factor (1 in the bf16 / fp32 builds: exact)
    if (base_at) {
#pragma unroll
        for (int i = 0; i < 12; ++i) g.v[i] = fmaf(br.v[i], inv_gs, g.v[i]);
    } else if (base) {
#pragma unroll
        for (int i = 0; i < 12; ++i) g.v[i] += br.v[i];
    }
    if (dx) g.store(dx + (size_t)row * D, lane);
    if (out3) g.store_split3(out3 + (size_t)row * SPLIT_A * D, lane, s3, hi3 != 0);   // fp32 split form: the next block's GELU' dgrad operand
    if (g_at) {
        Row12 gsc;
#pragma unroll
        for (int i = 0; i < 12; ++i) gsc.v[i] = g.v[i] * gs;
        gsc.store(g_at + (size_t)row * D, lane);
    }
    if (dmask_next) {
        float dm = 0.f;
        if (h_next && r >= 0) {
            Row12 hr;
            hr.load_at(h_next + (size_t)r * D, lane);
            LN_ROWPIN(hr);
            dm = dot12(g, hr);
        }
        if (lane == 0) dmask_next[row] = dm;
    }
}

int launch_ln_fwd(int precision, const float* x, const float* w, const float* b, void* out, float2* stats, int rows,
                  hipStream_t s, void* out3, int out3_f8) {
    const int grid = (rows + 3) / 4;
    if (dbg_skip(32)) return 0;
    if (precision == 0)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, x, w, b, (float*)out, stats, rows, (bf16*)out3, out3_f8);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, x, w, b, (bf16*)out, stats, rows, (bf16*)nullptr, 0);
    LAUNCH_CHECK();
    return 0;
}
int launch_ln_fwd_f32out(const float* x, const float* w, const float* b, float* out, int rows, hipStream_t s) {
    return launch_ln_fwd(0, x, w, b, out, nullptr, rows, s);
}
int launch_ln_bwd(int precision, const void* dy, const float* x, const float2* stats, const float* w, const float* base,
                  float* dx, int rows, void* g_at, const void* h_next, const int* dst_of_next, float* dmask_next,
                  float gs, hipStream_t s, void* out3, float s3, int out3_hi_only, const void* base_at) {
    if (dbg_skip(16)) return 0;
    if (precision == 0)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)dy, x, stats, w, base, dx,
                           rows, (float*)g_at, (const float*)h_next, dst_of_next, dmask_next, 1.0f, 1.0f, (bf16*)out3, s3, out3_hi_only, (const float*)nullptr);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<bf16>, dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16*)dy, x, stats, w, base, dx,
                           rows, (bf16*)g_at, (const bf16*)h_next, dst_of_next, dmask_next, gs, 1.0f / gs, (bf16*)nullptr, 1.0f, 0, (const bf16*)base_at);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// The adapter's own LayerNorm (tuning_config.ffn_adapter_layernorm_option "in" / "out"; reference models/dynamic_adapter.py:95-98,
// 121-122,132-133: nn.LayerNorm(768) with the default eps 1e-5, trainable, applied to the adapter's input or to its scaled output).
// Round 6: generic row kernels behind dyt_config::adapter_ln -- no shipped script uses the option, so nothing is fused.
//   adapter_ln_fwd:     out = LN(x) w + b (+ resid)   as OT (the 16-bit operand type: down-projection input; float: x_out = u + LN(up))
//   ln_param_grad:      per 32-row block:  partial[blk][0:768] = sum_t dy[t] xhat[t],  partial[blk][768:1536] = sum_t dy[t]   (x inv_gs)
//   ln_param_reduce:    out[i] += sum_blk partial[blk][i]   (fixed order)
// ------------------------------------------------------------------------------------------
constexpr float AD_LN_EPS = 1e-5f;
template <class OT>
__global__ __launch_bounds__(256) void adapter_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                             OT* __restrict__ out, float2* __restrict__ stats, const float* __restrict__ resid, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    Row12 xr, wr, br, rr;
    xr.load(x + (size_t)row * D, lane);
    wr.load(w, lane);
    br.load(b, lane);
    if (resid) rr.load(resid + (size_t)row * D, lane);
    xr.landed(); wr.landed(); br.landed();
    if (resid) rr.landed();
    const float mean = xr.sum() * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float d = xr.v[i] - mean; sq = fmaf(d, d, sq); }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / D) + AD_LN_EPS);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        xr.v[i] = (xr.v[i] - mean) * rstd * wr.v[i] + br.v[i];
        if (resid) xr.v[i] += rr.v[i];
    }
    xr.store(out + (size_t)row * D, lane);
    if (stats && lane == 0) stats[row] = make_float2(mean, rstd);
}
int launch_adapter_ln_fwd(int out_at_precision, const float* x, const float* w, const float* b, void* out, float2* stats, const float* resid,
                          int rows, hipStream_t s) {
    const int grid = (rows + 3) / 4;
    if (out_at_precision == 0)
        hipLaunchKernelGGL(adapter_ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, x, w, b, (float*)out, stats, resid, rows);
    else
        hipLaunchKernelGGL(adapter_ln_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, x, w, b, (bf16*)out, stats, resid, rows);
    LAUNCH_CHECK();
    return 0;
}

constexpr int LNPG_ROWS = 32;   // rows per workgroup
template <class AT>
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const AT* __restrict__ dy, const float* __restrict__ x, const float2* __restrict__ stats,
                                                            float* __restrict__ partial, int rows, float inv_gs) {
    __shared__ float red[4][2 * D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Row12 dg, db;
#pragma unroll
    for (int i = 0; i < 12; ++i) { dg.v[i] = 0.f; db.v[i] = 0.f; }
    const int r0 = blockIdx.x * LNPG_ROWS;
    for (int k = wave; k < LNPG_ROWS; k += 4) {
        const int r = r0 + k;
        if (r >= rows) break;
        Row12 g, xr;
        g.load_at(dy + (size_t)r * D, lane);
        xr.load(x + (size_t)r * D, lane);
        float2 st = stats[r];
        g.landed(); xr.landed();
        DYT_PIN2(st.x, st.y);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float gv = g.v[i] * inv_gs;
            db.v[i] += gv;
            dg.v[i] = fmaf(gv, (xr.v[i] - st.x) * st.y, dg.v[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][i * 256 + lane * 4 + e] = dg.v[4 * i + e];
            red[wave][D + i * 256 + lane * 4 + e] = db.v[4 * i + e];
        }
    __syncthreads();
    for (int c = tid; c < 2 * D; c += 256)
        partial[(size_t)blockIdx.x * (2 * D) + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * D) return;
    float t = 0.f;
    for (int p = 0; p < nparts; ++p) t += partial[(size_t)p * (2 * D) + i];
    out[i] += t;
}
// dgamma -> out[0:768], dbeta -> out[768:1536] (+=); dy carries the 16-bit gradient factor gs when precision != 0
int launch_ln_param_grad(int precision, const void* dy, const float* x, const float2* stats, float* partial, float* out, int rows, float gs, hipStream_t s) {
    const int nblk = (rows + LNPG_ROWS - 1) / LNPG_ROWS;
    if (precision == 0)
        hipLaunchKernelGGL(ln_param_grad_kernel<float>, dim3(nblk), dim3(256), 0, s, (const float*)dy, x, stats, partial, rows, 1.0f);
    else
        hipLaunchKernelGGL(ln_param_grad_kernel<bf16>, dim3(nblk), dim3(256), 0, s, (const bf16*)dy, x, stats, partial, rows, 1.0f / gs);
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * D + 255) / 256), dim3(256), 0, s, partial, nblk, out);
    LAUNCH_CHECK();
    return 0;
}
int64_t ln_param_grad_scratch_floats(int rows) { return (int64_t)((rows + LNPG_ROWS - 1) / LNPG_ROWS) * 2 * D; }

// the folded LayerNorm form's weight side (GemmArgs::ln_part), one wave per output feature n, from the fp32 weight (one rounding, like the
// plain 16-bit copy): Wf[n,:] = AT(gamma * W[n,:]), cs[n] = sum of the ROUNDED Wf[n,:] (what the matrix cores contract), bf[n] = bias[n] + <W[n,:], beta>
template <class AT>
__global__ __launch_bounds__(256) void ln_fold_w_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ bias, AT* __restrict__ Wf, float* __restrict__ cs,
                                                        float* __restrict__ bf, int N) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    Row12 wr, gr, br;
    wr.load(W + (size_t)n * D, lane);
    gr.load(gamma, lane);
    br.load(beta, lane);
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        t = fmaf(wr.v[i], br.v[i], t);
        wr.v[i] = to_f32(from_f32<AT>(wr.v[i] * gr.v[i]));
        s += wr.v[i];
    }
    s = wave_sum(s);
    t = wave_sum(t);
    wr.store(Wf + (size_t)n * D, lane);
    if (lane == 0) { cs[n] = s; bf[n] = bias[n] + t; }
}
int launch_ln_fold_w(int precision, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, void* Wfp, float* cs,
                     float* bf, int N, hipStream_t s) {
    if (precision == 0) { set_error("ln_fold_w: 16-bit modes only"); return -1; }
    hipLaunchKernelGGL(ln_fold_w_kernel<bf16>, dim3((N + 3) / 4), dim3(256), 0, s, W, gamma, beta, bias, (bf16*)Wf, cs, bf, N);
    LAUNCH_CHECK();
    if (Wfp) return launch_preshuffle_w(Wf, Wfp, N, D, s);
    return 0;
}

// ------------------------------------------------------------------------------------------
// token dispatcher: logits, (Gumbel-)sigmoid, hard threshold, per-image index compaction in LDS
// ------------------------------------------------------------------------------------------
// (1) logits for every patch token: one wave per token, whole chip
__global__ __launch_bounds__(256) void gate_logits_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ logit, int M) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= M) return;
    if (t % NT == 0) return;  // cls token is never gated
    Row12 wr, ur;
    wr.load(w, lane);
    ur.load(u + (size_t)t * D, lane);
    float bs = bias[0];
    wr.landed(); ur.landed(); DYT_PIN1(bs);
    const float l = dot12(ur, wr) + bs;
    if (lane == 0) logit[t] = l;
}
// (2) per image: (Gumbel-)sigmoid, hard threshold, ballot/popcount compaction of the kept-token list
__global__ __launch_bounds__(256) void gate_select_kernel(GateArgs a) {
    __shared__ int wave_cnt[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool keep = false;
    const int n = tid;
    if (n < NT) {
        const size_t t = (size_t)b * NT + n;
        if (n == 0) {
            keep = true;
            a.maskf[t] = 1.0f;
            a.soft[t] = 1.0f;
        } else {
            const float l = a.soft[t];  // gate_logits_kernel parked the logit here
            float z = l;
            if (a.training) {
                if (a.g1) {
                    z = (l + a.g1[(size_t)b * NP + n - 1] - a.g2[(size_t)b * NP + n - 1]) / a.tau;
                } else {
                    Philox ph(a.seed_dev ? *a.seed_dev : a.seed, a.subseq, t);
                    const float u = ph.u01(0);
                    z = (l + (logf(u) - log1pf(-u))) / a.tau;  // Gumbel - Gumbel ~ Logistic(0,1)
                }
            }
            const float sft = sigmoidf(z);
            keep = a.force_first > 0 ? n < a.force_first : sft > a.threshold;
            a.soft[t] = sft;
            a.maskf[t] = keep ? 1.0f : 0.0f;
            if (a.out_select) a.out_select[(size_t)b * a.out_stride + n - 1] = keep ? 1.0f : 0.0f;
            if (a.out_logits) a.out_logits[(size_t)b * a.out_stride + n - 1] = l;
        }
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (keep) a.keep_local[(size_t)b * NT + off + __popcll(bal & ((1ull << lane) - 1ull))] = n;
    if (tid == 0) a.counts[b] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

int launch_gate(const GateArgs& a, hipStream_t s) {
    const int M = a.batch * NT;
    hipLaunchKernelGGL(gate_logits_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a.u, a.w, a.b, a.soft, M);
    hipLaunchKernelGGL(gate_select_kernel, dim3(a.batch), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

template <class AT>
__global__ __launch_bounds__(256) void ln_gather_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                        const float* __restrict__ bb, const int* __restrict__ keep_local,
                                                        const int* __restrict__ counts, int* __restrict__ total,
                                                        const float* __restrict__ maskf, AT* __restrict__ out,
                                                        float2* __restrict__ stats, int* __restrict__ row_src,
                                                        int* __restrict__ dst_of, int batch, bf16* __restrict__ out3, int f8,
                                                        int* __restrict__ drop_src) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= batch * NT) return;
    const int b = slot / NT, j = slot - b * NT;
    const int cnt = counts[b];
    // exclusive prefix of the per-image counts, recomputed by every wave (B <= 1024 ints: cheaper than a
    // separate single-thread scan kernel on the critical path); wave 0 also publishes the grand total
    int off = 0;
    {
        const int lim = slot == 0 ? batch : b;
        int part = 0;
        for (int i = lane; i < lim; i += 64) part += counts[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (slot == 0) { if (lane == 0) { total[0] = part; if (drop_src) total[1] = batch * NT - part; } }
        else off = part;
    }
    // role 1: this wave owns TOKEN `slot`: a dropped token has no compact row
    if (maskf[slot] == 0.f) {
        if (lane == 0) dst_of[slot] = -1;
        if (drop_src) {   // ... and is entry (tokens before it - kept tokens before it) of the ascending list of dropped rows (gather_index_kernel's)
            int kept = 0;
            for (int i = lane; i < j; i += 64) kept += maskf[b * NT + i] != 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o, 64);
            if (lane == 0) drop_src[slot - off - kept] = slot;
        }
    }
    // role 2: this wave owns COMPACT slot j of image b
    if (j >= cnt) return;
    const int src = b * NT + keep_local[(size_t)b * NT + j];
    const int dst = off + j;
    Row12 xr, wr, br;
    xr.load(u + (size_t)src * D, lane);
    wr.load(w, lane);
    br.load(bb, lane);
    xr.landed(); wr.landed(); br.landed();
    const float2 st = ln_stats(xr);
#pragma unroll
    for (int i = 0; i < 12; ++i) xr.v[i] = (xr.v[i] - st.x) * st.y * wr.v[i] + br.v[i];
    if (out3) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (f8) store4_split_f8(out3 + (size_t)dst * SPLIT_A * D, D, i * 256 + lane * 4, xr.v[4 * i], xr.v[4 * i + 1], xr.v[4 * i + 2], xr.v[4 * i + 3]);
            else store4_split3(out3 + (size_t)dst * SPLIT_A * D + i * 256 + lane * 4, D, xr.v[4 * i], xr.v[4 * i + 1], xr.v[4 * i + 2], xr.v[4 * i + 3]);
    } else {
        xr.store(out + (size_t)dst * D, lane);
    }
    if (lane == 0) {
        stats[src] = st;
        row_src[dst] = src;
        dst_of[src] = dst;
    }
}

// token dispatcher index arrays without the gather (one thread per token): see ln_gather_kernel for the roles
__global__ __launch_bounds__(256) void gather_index_kernel(const int* __restrict__ keep_local, const int* __restrict__ counts,
                                                           int* __restrict__ total, const float* __restrict__ maskf,
                                                           int* __restrict__ row_src, int* __restrict__ dst_of, int batch,
                                                           int* __restrict__ drop_src) {
    __shared__ int off_s;
    __shared__ int wave_cnt[4];
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < 64) {   // exclusive prefix of the per-image counts (block 0 also publishes the grand total)
        const int lim = b == 0 ? batch : b;
        int part = 0;
        for (int i = j; i < lim; i += 64) part += counts[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (j == 0) {
            off_s = b == 0 ? 0 : part;
            if (b == 0) { total[0] = part; total[1] = batch * NT - part; }
        }
    }
    const int slot = b * NT + j;
    const bool dropped = j < NT && maskf[slot] == 0.f;
    const unsigned long long bal = __ballot(dropped);
    if ((j & 63) == 0) wave_cnt[j >> 6] = __popcll(bal);
    __syncthreads();
    if (j >= NT) return;
    if (dropped) {
        dst_of[slot] = -1;
        if (drop_src) {   // the dropped tokens, ascending: the row list of their up-projection launch (total[1] of them)
            int before = b * NT - off_s;   // dropped tokens of the images before this one
            for (int w = 0; w < (j >> 6); ++w) before += wave_cnt[w];
            drop_src[before + __popcll(bal & ((1ull << (j & 63)) - 1ull))] = slot;
        }
    }
    if (j < counts[b]) {
        const int src = b * NT + keep_local[(size_t)b * NT + j];
        row_src[off_s + j] = src;
        dst_of[src] = off_s + j;
    }
}
int launch_gather_index(const int* keep_local, const int* counts, int* total, const float* maskf, int* row_src, int* dst_of,
                        int batch, hipStream_t s, int* drop_src) {
    hipLaunchKernelGGL(gather_index_kernel, dim3(batch), dim3(256), 0, s, keep_local, counts, total, maskf, row_src, dst_of, batch, drop_src);
    LAUNCH_CHECK();
    return 0;
}

int launch_ln_gather(int precision, const float* u, const float* w, const float* b, const int* keep_local,
                     const int* counts, int* total, const float* maskf, void* out, float2* stats,
                     int* row_src, int* dst_of, int batch, hipStream_t s, void* out3, int out3_f8, int* drop_src) {
    const int grid = (batch * NT + 3) / 4;
    if (precision == 0)
        hipLaunchKernelGGL(ln_gather_kernel<float>, dim3(grid), dim3(256), 0, s, u, w, b, keep_local, counts, total,
                           maskf, (float*)out, stats, row_src, dst_of, batch, (bf16*)out3, out3_f8, drop_src);
    else
        hipLaunchKernelGGL(ln_gather_kernel<bf16>, dim3(grid), dim3(256), 0, s, u, w, b, keep_local, counts, total,
                           maskf, (bf16*)out, stats, row_src, dst_of, batch, (bf16*)nullptr, 0, drop_src);
    LAUNCH_CHECK();
    return 0;
}

template <class AT>
__global__ __launch_bounds__(256) void ln_cls_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                     const float* __restrict__ bb, AT* __restrict__ out,
                                                     float2* __restrict__ stats, AT* __restrict__ u_cls, int batch) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const size_t t = (size_t)b * NT;
    Row12 xr, wr, br;
    xr.load(u + t * D, lane);
    wr.load(w, lane);
    br.load(bb, lane);
    xr.landed(); wr.landed(); br.landed();
    if (u_cls) xr.store(u_cls + (size_t)b * D, lane);
    const float2 st = ln_stats(xr);
#pragma unroll
    for (int i = 0; i < 12; ++i) xr.v[i] = (xr.v[i] - st.x) * st.y * wr.v[i] + br.v[i];
    xr.store(out + (size_t)b * D, lane);
    if (lane == 0) stats[t] = st;
}
int launch_ln_cls(int precision, const float* u, const float* w, const float* b, void* out, float2* stats, void* u_cls,
                  int batch, hipStream_t s) {
    const int grid = (batch + 3) / 4;
    if (precision == 0)
        hipLaunchKernelGGL(ln_cls_kernel<float>, dim3(grid), dim3(256), 0, s, u, w, b, (float*)out, stats, (float*)u_cls, batch);
    else
        hipLaunchKernelGGL(ln_cls_kernel<bf16>, dim3(grid), dim3(256), 0, s, u, w, b, (bf16*)out, stats, (bf16*)u_cls, batch);
    LAUNCH_CHECK();
    return 0;
}
__global__ void cls_index_kernel(int* __restrict__ cls_rows, int batch) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < batch) cls_rows[b] = b * NT;
}
int launch_cls_index(int* cls_rows, int batch, hipStream_t s) {
    hipLaunchKernelGGL(cls_index_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, cls_rows, batch);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// patch embedding prologue
// ------------------------------------------------------------------------------------------
template <class AT>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, AT* __restrict__ out, int batch) {
    // out[(b*196 + py*14 + px)*768 + c*256 + i*16 + j] = img[b][c][py*16+i][px*16+j]; 4 consecutive j per thread
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)batch * NP * (D / 4);
    if (idx >= total) return;
    const int k4 = (int)(idx % (D / 4));
    const size_t row = idx / (D / 4);
    const int p = (int)(row % NP);
    const int b = (int)(row / NP);
    const int k = k4 * 4, c = k >> 8, i = (k >> 4) & 15, j = k & 15;
    const int py = p / 14, px = p - py * 14;
    const float4 v = *reinterpret_cast<const float4*>(img + (((size_t)b * 3 + c) * 224 + py * 16 + i) * 224 + px * 16 + j);
    store4(out + row * D + k, v.x, v.y, v.z, v.w);
}
int launch_im2col(int precision, const float* images, void* out, int batch, hipStream_t s) {
    const size_t total = (size_t)batch * NP * (D / 4);
    const int grid = (int)((total + 255) / 256);
    if (precision == 0) hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid), dim3(256), 0, s, images, (float*)out, batch);
    else hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(grid), dim3(256), 0, s, images, (bf16*)out, batch);
    LAUNCH_CHECK();
    return 0;
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x0, int batch) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= batch * D) return;
    const int b = idx / D, c = idx - b * D;
    x0[(size_t)b * NT * D + c] = cls[c] + pos[c];
}
int launch_cls_rows(const float* cls, const float* pos, float* x0, int batch, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((batch * D + 255) / 256), dim3(256), 0, s, cls, pos, x0, batch);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// dtype / layout conversion of weights
// ------------------------------------------------------------------------------------------
template <class AT>
__global__ void pad_convert_kernel(const float* __restrict__ src, AT* __restrict__ dst, int rows, int cols, int drows,
                                   int dcols, int transpose) {
    // dst[dr][dc] = transpose ? src[dc][dr] : src[dr][dc], zero outside the source
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)drows * dcols) return;
    const int dr = (int)(idx / dcols), dc = (int)(idx - (size_t)dr * dcols);
    float v = 0.f;
    if (transpose) { if (dc < rows && dr < cols) v = src[(size_t)dc * cols + dr]; }
    else { if (dr < rows && dc < cols) v = src[(size_t)dr * cols + dc]; }
    dst[idx] = from_f32<AT>(v);
}
static int pad_convert(int precision, const float* src, void* dst, int rows, int cols, int drows, int dcols, int tr,
                       hipStream_t s) {
    const size_t total = (size_t)drows * dcols;
    const int grid = (int)((total + 255) / 256);
    if (precision == 0)
        hipLaunchKernelGGL(pad_convert_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, rows, cols, drows, dcols, tr);
    else
        hipLaunchKernelGGL(pad_convert_kernel<bf16>, dim3(grid), dim3(256), 0, s, src, (bf16*)dst, rows, cols, drows, dcols, tr);
    LAUNCH_CHECK();
    return 0;
}
int launch_convert(int precision, const float* src, void* dst, int64_t n, hipStream_t s) {
    return pad_convert(precision, src, dst, 1, (int)n, 1, (int)n, 0, s);
}
int launch_transpose_convert(int precision, const float* src, void* dst, int rows, int cols, int dst_rows, int dst_cols,
                             hipStream_t s) {
    return pad_convert(precision, src, dst, rows, cols, dst_rows, dst_cols, 1, s);
}
int launch_pad_convert(int precision, const float* src, void* dst, int rows, int cols, int dst_rows, int dst_cols,
                       hipStream_t s) {
    return pad_convert(precision, src, dst, rows, cols, dst_rows, dst_cols, 0, s);
}

// ------------------------------------------------------------------------------------------
// head: final LayerNorm on the cls rows + Linear(768, C)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ nw,
                                                       const float* __restrict__ nb, const float* __restrict__ hw,
                                                       const float* __restrict__ hb, float* __restrict__ cls_n,
                                                       float2* __restrict__ stats, float* __restrict__ logits, int C) {
    __shared__ float row[D];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xp = x + (size_t)b * NT * D;
    float v[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { v[i] = xp[tid + 256 * i]; s += v[i]; }
    const float mean = block_sum256(s, red) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(block_sum256(q, red) * (1.0f / D) + LN_EPS);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid + 256 * i;
        const float y = (v[i] - mean) * rstd * nw[c] + nb[c];
        row[c] = y;
        cls_n[(size_t)b * D + c] = y;
    }
    if (tid == 0) stats[b] = make_float2(mean, rstd);
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
        float acc = 0.f;
        const float* wp = hw + (size_t)c * D;
#pragma unroll
        for (int i = 0; i < 12; ++i) acc = fmaf(row[lane + 64 * i], wp[lane + 64 * i], acc);
        acc = wave_sum(acc);
        if (lane == 0) logits[(size_t)b * C + c] = acc + hb[c];
    }
}
int launch_head_fwd(const float* x, const float* nw, const float* nb, const float* hw, const float* hb, float* cls_n,
                    float2* stats, float* logits, int batch, int C, hipStream_t s) {
    hipLaunchKernelGGL(head_fwd_kernel, dim3(batch), dim3(256), 0, s, x, nw, nb, hw, hb, cls_n, stats, logits, C);
    LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ dlogits, const float* __restrict__ x,
                                                          const float2* __restrict__ stats, const float* __restrict__ nw,
                                                          const float* __restrict__ hw, float* __restrict__ g, int C,
                                                          int g_stride) {
    __shared__ float red[4];
    __shared__ float dl[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) dl[c] = dlogits[(size_t)b * C + c];
    __syncthreads();
    const float2 st = stats[b];
    float dy[3], xh[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = tid + 256 * i;
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) acc = fmaf(dl[c], hw[(size_t)c * D + ch], acc);
        xh[i] = (x[(size_t)b * NT * D + ch] - st.x) * st.y;
        dy[i] = acc * nw[ch];
        s1 += dy[i];
        s2 = fmaf(dy[i], xh[i], s2);
    }
    s1 = block_sum256(s1, red) * (1.0f / D);
    s2 = block_sum256(s2, red) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < 3; ++i) g[(size_t)b * g_stride + tid + 256 * i] = st.y * (dy[i] - s1 - xh[i] * s2);
}
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* __restrict__ dlogits, const float* __restrict__ cls_n,
                                                          float* __restrict__ dW, float* __restrict__ db, int batch, int C) {
    // one class per workgroup; the batch loop is a latency chain (52 us at B=128 when every iteration waited for its own
    // loads), so the class's dlogits column goes to LDS first and the row loads of 8 images are issued together; the fmaf
    // chain per output keeps the image order
    __shared__ float dl[1024];
    const int c = blockIdx.x, tid = threadIdx.x;
    float acc[3] = {0.f, 0.f, 0.f};
    float sb = 0.f;
    for (int b0 = 0; b0 < batch; b0 += 1024) {
        const int nb = min(1024, batch - b0);
        __syncthreads();
        for (int b = tid; b < nb; b += 256) dl[b] = dlogits[(size_t)(b0 + b) * C + c];
        __syncthreads();
        int b = 0;
        for (; b + 8 <= nb; b += 8) {
            float v[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 3; ++i) v[u][i] = cls_n[(size_t)(b0 + b + u) * D + tid + 256 * i];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = dl[b + u];
                sb += d;
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = fmaf(d, v[u][i], acc[i]);
            }
        }
        for (; b < nb; ++b) {
            const float d = dl[b];
            sb += d;
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = fmaf(d, cls_n[(size_t)(b0 + b) * D + tid + 256 * i], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) dW[(size_t)c * D + tid + 256 * i] += acc[i];
    if (tid == 0) db[c] += sb;
}
int launch_head_bwd(const float* dlogits, const float* x, const float* cls_n, const float2* stats, const float* nw,
                    const float* hw, float* g, float* dWh, float* dbh, int batch, int C, int compact, hipStream_t s) {
    if (C > 1024) { set_error("head_bwd: num_classes %d > 1024 unsupported", C); return -1; }
    if (!compact) DYT_HIP_CHECK(hipMemsetAsync(g, 0, (size_t)batch * NT * D * sizeof(float), s));
    hipLaunchKernelGGL(head_bwd_dx_kernel, dim3(batch), dim3(256), 0, s, dlogits, x, stats, nw, hw, g, C,
                       compact ? D : NT * D);
    hipLaunchKernelGGL(head_bwd_dw_kernel, dim3(C), dim3(256), 0, s, dlogits, cls_n, dWh, dbh, batch, C);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// step loss (one workgroup; B x C is tiny)
// ------------------------------------------------------------------------------------------
// stage 1: one wave per image (4 per workgroup): log-softmax of both logit rows, their gradients, and the
// per-image CE / KL terms into part[b] = {ce_s, ce_t, kl}
__global__ __launch_bounds__(256) void loss_rows_kernel(LossArgs a, float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int C = a.C, B = a.batch;
    if (b >= B) return;
    const float invB = 1.0f / B;
    const float* ls = a.logits_s + (size_t)b * C;
    const float* lt = a.logits_t + (size_t)b * C;
    float ms = -INFINITY, mt = -INFINITY;
    for (int c = lane; c < C; c += 64) { ms = fmaxf(ms, ls[c]); mt = fmaxf(mt, lt[c]); }
    ms = wave_max(ms); mt = wave_max(mt);
    float ss = 0.f, st = 0.f;
    for (int c = lane; c < C; c += 64) { ss += expf(ls[c] - ms); st += expf(lt[c] - mt); }
    const float lse_s = ms + logf(wave_sum(ss)), lse_t = mt + logf(wave_sum(st));
    if (a.soft) {   // class-probability targets (LossArgs::soft): same form with the one-hot row replaced by t
        const float* tg = a.soft + (size_t)b * C;
        float tsum = 0.f;
        for (int c = lane; c < C; c += 64) tsum += tg[c];
        tsum = wave_sum(tsum);
        float klb = 0.f, ces = 0.f, cet = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float lps = ls[c] - lse_s, lpt = lt[c] - lse_t;
            const float ps = expf(lps), pt = expf(lpt), t = tg[c];
            klb += pt * (lpt - lps);
            ces -= t * lps; cet -= t * lpt;
            a.dlogits_s[(size_t)b * C + c] = ((ps * tsum - t) + (ps - pt)) * invB;
            a.dlogits_t[(size_t)b * C + c] = (pt * tsum - t) * invB;
        }
        klb = wave_sum(klb); ces = wave_sum(ces); cet = wave_sum(cet);
        if (lane == 0) { part[b * 4 + 0] = ces; part[b * 4 + 1] = cet; part[b * 4 + 2] = klb; }
        return;
    }
    const int y = (int)a.targets[b];
    float klb = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float lps = ls[c] - lse_s, lpt = lt[c] - lse_t;
        const float ps = expf(lps), pt = expf(lpt);
        klb += pt * (lpt - lps);
        const float oh = c == y ? 1.0f : 0.0f;
        a.dlogits_s[(size_t)b * C + c] = ((ps - oh) + (ps - pt)) * invB;
        a.dlogits_t[(size_t)b * C + c] = (pt - oh) * invB;
    }
    klb = wave_sum(klb);
    if (lane == 0) {
        part[b * 4 + 0] = -(ls[y] - lse_s);
        part[b * 4 + 1] = -(lt[y] - lse_t);
        part[b * 4 + 2] = klb;
    }
}
// stage 2: one workgroup reduces the per-image terms (fixed order) and the kept-token counts
__global__ __launch_bounds__(256) void loss_final_kernel(LossArgs a, const float* __restrict__ part) {
    __shared__ float red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.batch;
    float ce_s = 0.f, ce_t = 0.f, kl = 0.f, kept = 0.f;
    for (int b = tid; b < B; b += 256) { ce_s += part[b * 4]; ce_t += part[b * 4 + 1]; kl += part[b * 4 + 2]; }
    const int Bc = a.count_batch > 0 ? a.count_batch : B;   // images the gate statistics span
    if (a.counts)
        for (int i = tid; i < a.depth * Bc; i += 256) kept += (float)(a.counts[i] - 1);
    ce_s = wave_sum(ce_s); ce_t = wave_sum(ce_t); kl = wave_sum(kl); kept = wave_sum(kept);
    if (lane == 0) { red[wave][0] = ce_s; red[wave][1] = ce_t; red[wave][2] = kl; red[wave][3] = kept; }
    __syncthreads();
    if (tid == 0) {
        const float invB = 1.0f / B;
        const float base = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) * invB;
        const float teacher = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) * invB;
        const float klv = (red[0][2] + red[1][2] + red[2][2] + red[3][2]) * invB;
        float tok = 0.f, mean = 0.f, keptv = 0.f;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        if (a.counts) {
            keptv = red[0][3] + red[1][3] + red[2][3] + red[3][3];
            const float N = (float)a.depth * Bc * NP;
            mean = keptv / N;
            const float diff = mean - a.target_ratio;
            tok = diff * diff;
            d0 = a.loss_ratio * 2.0f * diff / N;
            if (a.token_minimal_weight > 0.f) {
                tok += a.token_minimal_weight * ((N - keptv) * fmaxf(a.token_minimal, 0.f) + keptv * fmaxf(a.token_minimal - 1.0f, 0.f));
                d1 = a.token_minimal > 0.f ? -a.loss_ratio * a.token_minimal_weight : 0.f;
                d2 = a.token_minimal > 1.f ? -a.loss_ratio * a.token_minimal_weight : 0.f;
            }
        }
        const float token_loss = a.loss_ratio * tok;
        a.out_losses[0] = base + token_loss + teacher + klv;
        a.out_losses[1] = base;
        a.out_losses[2] = token_loss;
        a.out_losses[3] = teacher;
        a.out_losses[4] = klv;
        a.out_losses[5] = mean;
        a.out_losses[6] = keptv;
        a.out_losses[7] = 0.f;
        a.dtok[0] = d0; a.dtok[1] = d1; a.dtok[2] = d2;
    }
}
int launch_loss(const LossArgs& a, hipStream_t s) {
    if (!a.scratch) { set_error("loss: scratch missing"); return -1; }
    hipLaunchKernelGGL(loss_rows_kernel, dim3((a.batch + 3) / 4), dim3(256), 0, s, a, a.scratch);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, a, a.scratch);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// AdamW over the flat trainable buffer
// ------------------------------------------------------------------------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float rsqrt_bc2, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
}
// ---- overflow-guarded update (the reference's GradScaler.step / update, misc.py:256-272: an update whose gradient holds inf / NaN is
// ---- skipped and counted; everything stays on the device: state = {updates applied, updates skipped, non-finite flag of this call, -}
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, int64_t n, int* __restrict__ state) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.0e38f);   // inf and NaN
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(state + 2, 1);
}
__global__ void adamw_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                     float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                     float gscale, const int* __restrict__ state) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || state[2] != 0) return;   // non-finite gradient somewhere: parameters and moments stay as they are
    const float step = (float)(state[0] + 1);
    const float bc1 = 1.0f - powf(b1, step), rsqrt_bc2 = 1.0f / sqrtf(1.0f - powf(b2, step));
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
}
__global__ void adamw_count_kernel(int* __restrict__ state) {
    if (threadIdx.x == 0) { if (state[2]) state[1] += 1; else state[0] += 1; }
}
int launch_adamw_guarded(float* p, const float* g, float* m, float* v, int64_t n, int* state, float lr, float b1, float b2, float eps,
                         float wd, float gscale, hipStream_t s) {
    if (hipMemsetAsync(state + 2, 0, sizeof(int), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return -2; }
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3((unsigned)std::min<int64_t>(256, (n + 255) / 256)), dim3(256), 0, s, g, n, state);
    hipLaunchKernelGGL(adamw_guarded_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, gscale, state);
    hipLaunchKernelGGL(adamw_count_kernel, dim3(1), dim3(64), 0, s, state);
    LAUNCH_CHECK();
    return 0;
}
__global__ void seed_set_kernel(uint64_t* seed_dev, uint64_t value, int advance) {
    if (threadIdx.x == 0) seed_dev[0] = advance ? seed_dev[0] + 1 : value;
}
int launch_seed_set(uint64_t* seed_dev, uint64_t value, hipStream_t s) {
    hipLaunchKernelGGL(seed_set_kernel, dim3(1), dim3(64), 0, s, seed_dev, value, 0);
    LAUNCH_CHECK();
    return 0;
}
int launch_seed_advance(uint64_t* seed_dev, hipStream_t s) {
    hipLaunchKernelGGL(seed_set_kernel, dim3(1), dim3(64), 0, s, seed_dev, 0ull, 1);
    LAUNCH_CHECK();
    return 0;
}

// ---- global-norm clipping of the flat gradient (engine_finetune.py:74 -> misc.py:262-266 clip_grad_norm_) ----
__global__ __launch_bounds__(256) void sqsum_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
    __shared__ float ws[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ part, int nparts,
                                                         float max_norm, float pre_scale, float* __restrict__ norm_out) {
    __shared__ float coef_s;
    if (threadIdx.x < 64) {   // every block re-derives the (deterministic, fixed-order) total: no second launch, no atomics
        float acc = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
        acc = wave_sum(acc);
        if (threadIdx.x == 0) {
            const float norm = sqrtf(acc) * fabsf(pre_scale);
            coef_s = fminf(1.0f, max_norm / (norm + 1e-6f));
            if (blockIdx.x == 0 && norm_out) norm_out[0] = norm;
        }
    }
    __syncthreads();
    const float coef = coef_s;
    if (coef >= 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] *= coef;
}
int launch_clip_grad_norm(float* g, int64_t n, float max_norm, float pre_scale, float* scratch, float* norm_out, hipStream_t s) {
    const int nparts = 256;
    hipLaunchKernelGGL(sqsum_kernel, dim3(nparts), dim3(256), 0, s, g, n, scratch);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(nparts), dim3(256), 0, s, g, n, scratch, nparts, max_norm, pre_scale, norm_out);
    LAUNCH_CHECK();
    return 0;
}

int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                 float wd, float bc1, float bc2, float gscale, hipStream_t s) {
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps,
                       wd, bc1, 1.0f / sqrtf(bc2), gscale);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// backward prep of a block: AT copy of g, gathered MLP gradient rows, <g, h> per token
// ------------------------------------------------------------------------------------------
template <class AT>
__global__ __launch_bounds__(256) void bwd_prep_kernel(const float* __restrict__ g, const AT* __restrict__ h,
                                                       const int* __restrict__ dst_of, const float* __restrict__ row_mask,
                                                       AT* __restrict__ g_at, AT* __restrict__ dH,
                                                       float* __restrict__ dmask, int M, float gs) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= M) return;
    Row12 gr;
    gr.load(g + (size_t)t * D, lane);
    int r = dst_of ? dst_of[t] : t;
    float mk = row_mask ? row_mask[t] : 1.0f;
    gr.landed(); DYT_PIN2(r, mk);
    if (g_at) {
        Row12 gsc;
#pragma unroll
        for (int i = 0; i < 12; ++i) gsc.v[i] = gr.v[i] * gs;
        gsc.store(g_at + (size_t)t * D, lane);
    }
    float dm = 0.f;
    if (r >= 0) {
        if (h) {
            Row12 hr;
            hr.load_at(h + (size_t)r * D, lane);
            hr.landed();
            dm = dot12(gr, hr);
        }
        if (dH) {
            if (row_mask) {
#pragma unroll
                for (int i = 0; i < 12; ++i) gr.v[i] *= mk;
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) gr.v[i] *= gs;
            gr.store(dH + (size_t)r * D, lane);
        }
    }
    if (dmask && lane == 0) dmask[t] = dm;
}
int launch_bwd_prep(int precision, const BwdPrepArgs& a, hipStream_t s) {
    const int grid = (a.M + 3) / 4;
    if (precision == 0)
        hipLaunchKernelGGL(bwd_prep_kernel<float>, dim3(grid), dim3(256), 0, s, a.g, (const float*)a.h, a.dst_of, a.row_mask,
                           (float*)a.g_at, (float*)a.dH, a.dmask, a.M, 1.0f);
    else
        hipLaunchKernelGGL(bwd_prep_kernel<bf16>, dim3(grid), dim3(256), 0, s, a.g, (const bf16*)a.h, a.dst_of, a.row_mask,
                           (bf16*)a.g_at, (bf16*)a.dH, a.dmask, a.M, a.gs);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// per-token tail of a block's backward
// ------------------------------------------------------------------------------------------
constexpr int TOK_PER_BLOCK = 32;

template <class AT>
__global__ __launch_bounds__(256) void tok_bwd_kernel(TokBwdArgs a) {
    __shared__ float red[4][D + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Row12 wg, ln2w, dwg;
    float dbg = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) dwg.v[i] = 0.f;
    if (a.gate_w) { wg.load(a.gate_w, lane); TK_ROWPIN(wg); }
    if (a.dA2) { ln2w.load(a.ln2_w, lane); TK_ROWPIN(ln2w); }
    const int t0 = blockIdx.x * TOK_PER_BLOCK;
    // compact rows of this wave's tokens, one per lane, loaded up front: the dA2 row of a token then goes out WITH its other rows instead of
    // behind the drain that delivers its index (round 6: one memory latency per token instead of two)
    int rv = -1;
    if (a.dA2 && a.write_du) {
        const int tt = t0 + wave + 4 * lane;
        if (lane < TOK_PER_BLOCK / 4 && tt < a.M) {
            if (a.g_cls) { const int bb = tt / NT; rv = (tt - bb * NT == 0) ? bb : -1; }
            else rv = a.dst_of ? a.dst_of[tt] : tt;
        }
    }
    for (int k = wave; k < TOK_PER_BLOCK; k += 4) {
        const int t = t0 + k;
        if (t >= a.M) break;
        const int b = t / NT, n = t - b * NT;
        // all loads of the token first, then ONE full drain that names every one of them (DYT_PIN*, dyt_common.h): the previous
        // iteration's stores, the row loads (misses) and the per-token scalars share the counter
        Row12 du, ur, e, dy;
        const bool need_u = (a.dA2 && (!a.g_cls || n == 0)) || a.gate_w;   // cls-only tail without a gate: only the cls rows have an LN2 backward
        const bool gate = a.gate_w && n >= 1;
        const bool cat = gate && a.cat_dact && a.dmask;      // the saved MLP output includes the adapter's weight term (see TokBwdArgs)
        if (need_u) ur.load_nt(a.u + (size_t)t * D, lane);   // saved u of the forward pass: last use
        const bool du16 = a.du_in_at && a.write_du && !a.g_cls;   // the incoming gradient as the 16-bit gs-scaled copy (TokBwdArgs::du_in_at)
        if (du16) du.load_at(reinterpret_cast<const AT*>(a.du_in_at) + (size_t)t * D, lane);
        else if (a.write_du && !a.g_cls) du.load(a.du + (size_t)t * D, lane);
        else if (a.write_du && n == 0) du.load(a.g_cls + (size_t)b * D, lane);
        else {
#pragma unroll
            for (int i = 0; i < 12; ++i) du.v[i] = 0.f;
        }
        if (a.dad) e.load_at(reinterpret_cast<const AT*>(a.dad) + (size_t)t * D, lane);
        const int r = __builtin_amdgcn_readlane(rv, k >> 2);   // (k >> 2 is wave-uniform: k = wave + 4 j)
        if (r >= 0) dy.load_at(reinterpret_cast<const AT*>(a.dA2) + (size_t)r * D, lane);
        float2 st2 = make_float2(0.f, 1.f);
        if (a.dA2 && a.write_du) st2 = a.stats2[t];
        float ext = 0.f, sf = 0.f, dmk = 0.f, dlg = 0.f;
        float cd = 0.f, cz = 0.f;   // this lane's column of the token's d_act / ddz rows (saved h includes the adapter, see TokBwdArgs)
        if (gate) {
            const size_t oi = (size_t)b * a.out_stride + n - 1;
            if (a.dtoken_select) ext = a.dtoken_select[oi];
            else if (a.dtok) ext = a.dtok[0] + (a.maskf[t] != 0.f ? a.dtok[2] : a.dtok[1]);
            sf = a.soft[t];
            if (a.dmask) dmk = a.dmask[t];
            if (a.dtoken_logits) dlg = a.dtoken_logits[oi];
            if (cat) {
                cd = to_f32(reinterpret_cast<const AT*>(a.cat_dact)[(size_t)t * RP + lane]);
                cz = to_f32(reinterpret_cast<const AT*>(a.cat_ddz)[(size_t)t * RP + lane]);
            }
        }
        if (need_u) TK_ROWPIN(ur);
        TK_ROWPIN(du);
        if (a.dad) TK_ROWPIN(e);
        if (r >= 0) TK_ROWPIN(dy);
        TK_SCALPIN3(st2.x, st2.y, ext);
        TK_SCALPIN3(sf, dmk, dlg);
        if (cat) {
            DYT_PIN2(cd, cz);
            if (a.maskf[t] != 0.f) dmk -= wave_sum(cd * cz) * a.cat_ddz_scale;   // dropped tokens have dmask = 0 and no saved h
        }
        const float bs = a.branch_scale ? a.branch_scale[b] : 1.0f;   // stochastic depth on the MLP branch (uniform per image)
        dmk *= bs;
        if (du16) {
#pragma unroll
            for (int i = 0; i < 12; ++i) du.v[i] *= a.inv_gs;
        }
        if (a.dad) {
#pragma unroll
            for (int i = 0; i < 12; ++i) du.v[i] += e.v[i] * a.inv_gs;
        }
        if (a.dA2 && a.write_du) {
            if (r >= 0) {
                ln_bwd_row(dy, ur, ln2w, st2);
                const float ds = a.inv_gs * bs;
#pragma unroll
                for (int i = 0; i < 12; ++i) du.v[i] += dy.v[i] * ds;
            }
        }
        if (gate) {
            float dlogit = (dmk + ext) * sf * (1.0f - sf);
            if (a.training) dlogit /= a.tau;
            dlogit += dlg;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                du.v[i] = fmaf(dlogit, wg.v[i], du.v[i]);
                dwg.v[i] = fmaf(dlogit, ur.v[i], dwg.v[i]);
            }
            dbg += dlogit;
        }
        if (a.write_du) {
            if (a.du) du.store(a.du + (size_t)t * D, lane);
            if (a.du3) du.store_split3(reinterpret_cast<bf16*>(a.du3) + (size_t)t * SPLIT_A * D, lane, a.du3_scale, a.du3_hi_only);   // proj dgrad operand (fp32 split form)
            if (a.du_at) {
                Row12 dsc;
#pragma unroll
                for (int i = 0; i < 12; ++i) dsc.v[i] = du.v[i] * a.gs;
                dsc.store(reinterpret_cast<AT*>(a.du_at) + (size_t)t * D, lane);
            }
        }
    }
    if (a.gate_w) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][i * 256 + lane * 4 + e] = dwg.v[4 * i + e];
        if (lane == 0) red[wave][D] = dbg;  // dbg is lane-uniform
        __syncthreads();
        for (int c = tid; c < D + 1; c += 256)
            a.partial[(size_t)blockIdx.x * (D + 1) + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}
int launch_tok_bwd(int precision, const TokBwdArgs& a, int* nblocks_out, hipStream_t s) {
    const int grid = (a.M + TOK_PER_BLOCK - 1) / TOK_PER_BLOCK;
    if (nblocks_out) *nblocks_out = grid;
    if (dbg_skip(8)) return 0;
    if (precision == 0) hipLaunchKernelGGL(tok_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(tok_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// stochastic depth (timm DropPath; reference models/vision_transformer_IN21K.py:121,131,148,159 with dpr = linspace(0, rate, depth), :285)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void drop_path_draw_kernel(float* __restrict__ scales, int depth, int batch, float rate, uint64_t seed,
                                                             const uint64_t* __restrict__ seed_dev, uint64_t subseq_base) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // (branch, l, b)
    if (i >= 2 * depth * batch) return;
    const int b = i % batch, l = (i / batch) % depth, branch = i / (batch * depth);
    const float drop = depth > 1 ? rate * (float)l / (float)(depth - 1) : 0.f;
    float v = 1.0f;
    if (drop > 0.f) {
        const float keep = 1.0f - drop;
        Philox ph(seed_dev ? *seed_dev : seed, subseq_base + (uint64_t)(2 * l + branch), (uint64_t)b);
        v = ph.u01(0) < keep ? 1.0f / keep : 0.0f;
    }
    scales[i] = v;
}
int launch_drop_path_draw(float* scales, int depth, int batch, float rate, uint64_t seed, const uint64_t* seed_dev, uint64_t subseq_base, hipStream_t s) {
    hipLaunchKernelGGL(drop_path_draw_kernel, dim3((2 * depth * batch + 255) / 256), dim3(256), 0, s, scales, depth, batch, rate, seed, seed_dev, subseq_base);
    LAUNCH_CHECK();
    return 0;
}
template <class AT>
__global__ __launch_bounds__(256) void scale_rows_kernel(AT* __restrict__ x, const float* __restrict__ scale, int M, int ld) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;   // 4 consecutive elements of a row (ld % 4 == 0)
    if (i >= (size_t)M * ld) return;
    const float sc = scale[(int)(i / ld) / NT];
    float v[4];
    load4(x + i, v);
    store4(x + i, v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
}
int launch_scale_rows(int precision, void* x, const float* scale, int M, int ld, hipStream_t s) {
    const unsigned grid = (unsigned)(((size_t)M * ld / 4 + 255) / 256);
    if (precision == 0) hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)x, scale, M, ld);
    else hipLaunchKernelGGL(scale_rows_kernel<bf16>, dim3(grid), dim3(256), 0, s, (bf16*)x, scale, M, ld);
    LAUNCH_CHECK();
    return 0;
}

// 32 outputs x 8 partial-groups per workgroup; fixed summation order (deterministic)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nparts, int stride,
                                                              float* __restrict__ out, int n, float alpha) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    float acc = 0.f;
    if (i < n) {
        const int per = (nparts + 7) / 8;
        const int p0 = grp * per, p1 = min(nparts, p0 + per);
#pragma unroll 4
        for (int p = p0; p < p1; ++p) acc += partial[(size_t)p * stride + i];
    }
    red[grp][o] = acc;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][o];
        out[i] += alpha * t;
    }
}
int launch_reduce_partials(const float* partial, int nparts, int stride, float* out, int n, float alpha, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 31) / 32), dim3(256), 0, s, partial, nparts, stride, out, n, alpha);
    LAUNCH_CHECK();
    return 0;
}

// the same for several (partial, out) pairs in one launch: blockIdx.y selects the pair
struct TokReduceBatch { const float* partial[ReduceQueue::MAX_TOK]; float* out[ReduceQueue::MAX_TOK]; int nparts[ReduceQueue::MAX_TOK]; };
__global__ __launch_bounds__(256) void reduce_partials_batch_kernel(TokReduceBatch b, int stride, int n) {
    __shared__ float red[8][32];
    const float* __restrict__ partial = b.partial[blockIdx.y];
    const int nparts = b.nparts[blockIdx.y];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    float acc = 0.f;
    if (i < n) {
        const int per = (nparts + 7) / 8;
        const int p0 = grp * per, p1 = min(nparts, p0 + per);
#pragma unroll 4
        for (int p = p0; p < p1; ++p) acc += partial[(size_t)p * stride + i];
    }
    red[grp][o] = acc;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][o];
        b.out[blockIdx.y][i] += t;
    }
}
int queue_tok_reduce(ReduceQueue& q, const float* partial, int nparts, float* out) {
    if (q.n_tok >= ReduceQueue::MAX_TOK) { set_error("reduce queue full"); return -1; }
    q.tok_partial[q.n_tok] = partial; q.tok_out[q.n_tok] = out; q.tok_nparts[q.n_tok] = nparts; ++q.n_tok;
    return 0;
}

// ------------------------------------------------------------------------------------------
// adapter weight gradients:  C[c][j] = sum_m X[m][c] Y[m][j]   (+ column sums of X in j = 64)
// ------------------------------------------------------------------------------------------
constexpr int WG_J = 80;        // 64 adapter columns + the ones column (+ pad)
constexpr int WG_CHUNK = 512;   // minimum tokens per workgroup (the partial buffers are sized for M / 512 chunks)

// bf16: MFMA 32x32x16 with the token dimension as K; tiles are transposed while staged to LDS.
// 64 tokens per step; the next step's rows are prefetched into registers while the current step's
// fragments are read and multiplied (the loads were fully exposed before: 1.3 TB/s).  The column sums
// of X (-> partial[..][c][64]) and of Y (wave 0 of the first channel block, -> partial[..][768][j]) are
// taken from the same fragments on the vector ALU.
constexpr int WG_ROWS = D + 8;  // partial rows per chunk: 768 channels + 1 row of Y column sums (+pad)
struct WgSrc { const void* X; const void* Y; float* partial; float xscale, yscale; };   // x / yscale: fp32 inputs of the half-product form are multiplied by these before the 16-bit conversion
struct WgPair { WgSrc p[2]; };   // blockIdx.z selects the product: both adapter weight gradients of a block are ONE launch
// IT = float: the same kernel on fp32 inputs, converted (times a power of two that keeps gradient-sized values out of the 16-bit
// subnormals) as they are loaded -- the one-part gradient products of the "fp16x3f" form (wgrad_f32_kernel runs at the fp32-MFMA rate)
template <class IT>
__device__ __forceinline__ bf16x8 wg_load8(const IT* p, float scale) {
    if constexpr (sizeof(IT) == 2) {
        return *reinterpret_cast<const bf16x8*>(p);
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p) * scale, b = *reinterpret_cast<const f32x4*>(p + 4) * scale;
        bf16x8 v = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
        return v;
    }
}
template <class IT>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(WgPair src, int M, int chunk) {
    const IT* __restrict__ X = static_cast<const IT*>(src.p[blockIdx.z].X);
    const IT* __restrict__ Y = static_cast<const IT*>(src.p[blockIdx.z].Y);
    const float xscale = src.p[blockIdx.z].xscale, yscale = src.p[blockIdx.z].yscale;
    float* __restrict__ partial = src.p[blockIdx.z].partial;
    constexpr int TS = 64;        // tokens per step
    // LDS rows of 64 tokens (128 B) with the eight 16-B chunks of row r XOR-swizzled by s(r) = ((r >> 3) ^ r) & 7.  Round 6: the round 1-5 layout
    // (144-B rows, no swizzle) kept the ds_read_b128 fragment reads conflict-free but put all 16 channel-chunk lanes of a ds_write_b32
    // lane group on ONE bank (8 rows = 288 words = 0 mod 32): 16-way conflicts on the 24 transposing writes of every step, i.e. the kernel
    // ran on LDS-write cycles (37 us for 84 MB).  With the swizzle the reads stay conflict-free and the writes are 2-way (X) / conflict-free
    // (Y); enumerated for every lane group of both instructions in tools/probes/r6/wgrad_swizzle.py.
    constexpr int LDT = TS;
    auto swz = [](int r) { return ((r >> 3) ^ r) & 7; };
    __shared__ __attribute__((aligned(16))) bf16 Xt[128 * LDT];
    __shared__ __attribute__((aligned(16))) bf16 Yt[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * 128;
    const int m0 = blockIdx.y * chunk;
    const int mend = min(m0 + chunk, M);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float xs = 0.f, ys[2] = {0.f, 0.f};   // column sums of X (own channel) / Y (own columns) over this lane's tokens
    const bool do_ysum = blockIdx.x == 0 && wave == 0;

    // loader roles: X tile = 32 token pairs x 16 channel chunks = 512 tasks (2 / thread); Y = 32 x 8 (1 / thread)
    const int xp0 = tid >> 4, xc = tid & 15;
    const int yp = tid >> 3, yc = tid & 7;
    bf16x8 xr[2][2], yr[2];
    auto zero = [](bf16x8& v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)0.f;
    };
    auto load_step = [&](int tb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = tb + 2 * (xp0 + 16 * h);
            zero(xr[h][0]); zero(xr[h][1]);
            if (t < mend) xr[h][0] = wg_load8<IT>(X + (size_t)t * D + c0 + xc * 8, xscale);
            if (t + 1 < mend) xr[h][1] = wg_load8<IT>(X + (size_t)(t + 1) * D + c0 + xc * 8, xscale);
        }
        const int ty = tb + 2 * yp;
        zero(yr[0]); zero(yr[1]);
        if (ty < mend) yr[0] = wg_load8<IT>(Y + (size_t)ty * RP + yc * 8, yscale);
        if (ty + 1 < mend) yr[1] = wg_load8<IT>(Y + (size_t)(ty + 1) * RP + yc * 8, yscale);
    };
    if (m0 < mend) load_step(m0);
    for (int tb = m0; tb < mend; tb += TS) {
        __syncthreads();  // previous step's fragment reads are done
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x2 pv = {xr[h][0][i], xr[h][1][i]};
                const int tk = 2 * (xp0 + 16 * h);   // token (pair) column of row xc * 8 + i: chunk tk / 8 goes to slot (tk / 8) ^ s(row), s = (xc ^ i) & 7
                *reinterpret_cast<bf16x2*>(&Xt[(xc * 8 + i) * LDT + ((((tk >> 3) ^ xc ^ i) & 7) << 3) + (tk & 7)]) = pv;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bf16x2 pv = {yr[0][i], yr[1][i]};
            const int tk = 2 * yp;
            *reinterpret_cast<bf16x2*>(&Yt[(yc * 8 + i) * LDT + ((((tk >> 3) ^ yc ^ i) & 7) << 3) + (tk & 7)]) = pv;
        }
        __syncthreads();
        if (tb + TS < mend) load_step(tb + TS);  // in flight during the reads + MFMAs below
        const int lr = lane & 31, lk = lane >> 5;
        auto sum8 = [](const bf16x8& v) {   // sum of the 8 operands in fp32
#ifdef DYT_FP16
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += (float)v[i];
            return t;
#else
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // bf16 -> fp32 is a 16-bit shift
            const u32x4 u = __builtin_bit_cast(u32x4, v);
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) t += __builtin_bit_cast(float, u[i] << 16) + __builtin_bit_cast(float, u[i] & 0xffff0000u);
            return t;
#endif
        };
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {  // four 16-token k blocks; lane = (row lr, tokens kk*16 + lk*8 .. +7)
            const int xrow = wave * 32 + lr;
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&Xt[xrow * LDT + (((kk * 2 + lk) ^ swz(xrow)) << 3)]);
            bf16x8 yf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) yf[j] = *reinterpret_cast<const bf16x8*>(&Yt[(j * 32 + lr) * LDT + (((kk * 2 + lk) ^ swz(j * 32 + lr)) << 3)]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = DYT_MFMA_32x32x16(xf, yf[j], acc[j]);
            xs += sum8(xf);
            if (do_ysum) { ys[0] += sum8(yf[0]); ys[1] += sum8(yf[1]); }
        }
    }
    // D register i of block j: row = (i / 4) * 8 + (lane >> 5) * 4 + (i % 4) -> channel, col = lane & 31 -> column j * 32 + col
    float* pp = partial + (size_t)blockIdx.y * WG_ROWS * WG_J;
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + wave * 32 + (i >> 2) * 8 + lk * 4 + (i & 3);
            pp[(size_t)c * WG_J + j * 32 + lr] = acc[j][i];
        }
    xs += __shfl_xor(xs, 32, 64);   // the two token groups of a channel
    if (lk == 0) pp[(size_t)(c0 + wave * 32 + lr) * WG_J + 64] = xs;
    if (do_ysum) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = ys[j] + __shfl_xor(ys[j], 32, 64);
            if (lk == 0) pp[(size_t)D * WG_J + j * 32 + lr] = v;
        }
    }
}

// fp32 exact variant on the matrix cores: v_mfma_f32_32x32x2_f32 with the TOKEN dimension as k.  One fp32 per lane and
// operand: lane l supplies (row l % 32, k = l / 32), so an A operand is X[token m + l/32][channel] and a B operand
// Y[token m + l/32][column] -- both straight from global memory, no transposition and no LDS staging.  A lane loads a
// float4 of X (4 consecutive channels) and a float2 of Y per token pair and feeds register t / u to MFMA (t, u): that
// MFMA owns channels c0 + 4*row + t and columns 2*col + u (only the ownership is permuted).  A wave covers 128 channels x
// 64 columns (8 MFMAs, 128 accumulator registers) and every 4th token pair of the chunk; the 4 waves of a workgroup are
// summed through LDS in wave order (deterministic).  Column sums of X / Y ride along on the vector ALU.
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgPair src, int M, int chunk) {
    const float* __restrict__ X = static_cast<const float*>(src.p[blockIdx.z].X);
    const float* __restrict__ Y = static_cast<const float*>(src.p[blockIdx.z].Y);
    float* __restrict__ partial = src.p[blockIdx.z].partial;
    __shared__ float red[134 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * 128;
    const int m0 = blockIdx.y * chunk, mend = min(m0 + chunk, M);
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
    float xs[4] = {0.f, 0.f, 0.f, 0.f}, ys[2] = {0.f, 0.f};
    const float* xp = X + (size_t)c0 + 4 * lr;
    const float* yp = Y + 2 * lr;
    constexpr int PF = 4;   // token pairs in flight per wave
    f32x4 xa[PF];
    f32x2 yb[PF];
    auto load_pair = [&](int slot, int m) {   // m = first token of the pair; this lane reads token m + lk
        const int t = m + lk;
        xa[slot] = f32x4{0.f, 0.f, 0.f, 0.f};
        yb[slot] = f32x2{0.f, 0.f};
        if (t < mend) {
            xa[slot] = *reinterpret_cast<const f32x4*>(xp + (size_t)t * D);
            yb[slot] = *reinterpret_cast<const f32x2*>(yp + (size_t)t * RP);
        }
    };
    int m = m0 + 2 * wave;   // this wave's token pairs: m, m + 8, m + 16, ...
#pragma unroll
    for (int q = 0; q < PF; ++q) load_pair(q, m + 8 * q);
    for (; m < mend; m += 8 * PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const f32x4 a = xa[q];
            const f32x2 b = yb[q];
            load_pair(q, m + 8 * (PF + q));   // refill the slot (zeros beyond the chunk)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[0], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[1], acc[t][1], 0, 0, 0);
                xs[t] += a[t];
            }
            ys[0] += b[0]; ys[1] += b[1];
        }
    }
    // ---- sum the 4 waves in wave order through LDS: element k of lane l lives at red[k * 64 + l] ----
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float* q = &red[((t * 2 + u) * 16 + i) * 64 + lane];
                        if (w == 0) *q = acc[t][u][i]; else if (w < 3) *q += acc[t][u][i]; else acc[t][u][i] += *q;
                    }
#pragma unroll
            for (int t = 0; t < 4; ++t) { float* q = &red[(128 + t) * 64 + lane]; if (w == 0) *q = xs[t]; else if (w < 3) *q += xs[t]; else xs[t] += *q; }
#pragma unroll
            for (int u = 0; u < 2; ++u) { float* q = &red[(132 + u) * 64 + lane]; if (w == 0) *q = ys[u]; else if (w < 3) *q += ys[u]; else ys[u] += *q; }
        }
        __syncthreads();
    }
    if (wave != 3) return;
    // D register i of MFMA (t, u): row = (i / 4) * 8 + lk * 4 + (i % 4), col = lr -> channel c0 + 4 * row + t, column 2 * lr + u
    float* pp = partial + (size_t)blockIdx.y * WG_ROWS * WG_J;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i >> 2) * 8 + lk * 4 + (i & 3);
            const int c = c0 + 4 * row + t;
            *reinterpret_cast<f32x2*>(&pp[(size_t)c * WG_J + 2 * lr]) = f32x2{acc[t][0][i], acc[t][1][i]};
        }
    // column sums: the two token parities (lanes l and l ^ 32) are still separate
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float v = xs[t] + __shfl_xor(xs[t], 32, 64);
        if (lk == 0) pp[(size_t)(c0 + 4 * lr + t) * WG_J + 64] = v;
    }
    if (blockIdx.x == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float v = ys[u] + __shfl_xor(ys[u], 32, 64);
            if (lk == 0) pp[(size_t)D * WG_J + 2 * lr + u] = v;
        }
    }
}

struct WgOut {
    const float* partial; float* out_w; int sc, sj; float alpha; float* out_xsum; float alpha_x; float* out_ysum; float alpha_y;
};
struct WgOutPair { WgOut p[2]; };
__global__ void wgrad_reduce_kernel(WgOutPair outs, int nchunks, int r) {
    const WgOut& o = outs.p[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (D + 1) * (r + 1)) return;
    const int c = idx / (r + 1), j = idx - c * (r + 1);
    const int col = j < r ? j : 64;
    if (c == D && j == r) return;
    float acc = 0.f;
    for (int p = 0; p < nchunks; ++p) acc += o.partial[((size_t)p * WG_ROWS + c) * WG_J + col];
    if (c == D) { if (o.out_ysum) o.out_ysum[j] += o.alpha_y * acc; }
    else if (j < r) o.out_w[(size_t)c * o.sc + (size_t)j * o.sj] += o.alpha * acc;
    else if (o.out_xsum) o.out_xsum[c] += o.alpha_x * acc;
}

// the reduce of several products (each with its own partial buffer and chunk count) in one launch: blockIdx.y selects the product
struct WgReduceBatch { WgReduceDesc e[ReduceQueue::MAX_WG]; };
__global__ void wgrad_reduce_batch_kernel(WgReduceBatch b, int r) {
    const WgReduceDesc& o = b.e[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (D + 1) * (r + 1)) return;
    const int c = idx / (r + 1), j = idx - c * (r + 1);
    const int col = j < r ? j : 64;
    if (c == D && j == r) return;
    float acc = 0.f;
    for (int p = 0; p < o.nchunks; ++p) acc += o.partial[((size_t)p * WG_ROWS + c) * WG_J + col];
    if (c == D) { if (o.out_ysum) o.out_ysum[j] += o.alpha_y * acc; }
    else if (j < r) o.out_w[(size_t)c * o.sc + (size_t)j * o.sj] += o.alpha * acc;
    else if (o.out_xsum) o.out_xsum[c] += o.alpha_x * acc;
}
int flush_reductions(ReduceQueue& q, hipStream_t s) {
    if (q.n_wg > 0) {
        WgReduceBatch b;
        for (int i = 0; i < q.n_wg; ++i) b.e[i] = q.wg[i];
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(((D + 1) * (q.r + 1) + 255) / 256, q.n_wg), dim3(256), 0, s, b, q.r);
        q.n_wg = 0;
    }
    if (q.n_tok > 0) {
        TokReduceBatch b;
        for (int i = 0; i < q.n_tok; ++i) { b.partial[i] = q.tok_partial[i]; b.out[i] = q.tok_out[i]; b.nparts[i] = q.tok_nparts[i]; }
        hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3((D + 1 + 31) / 32, q.n_tok), dim3(256), 0, s, b, D + 1, D + 1);
        q.n_tok = 0;
    }
    LAUNCH_CHECK();
    return 0;
}

// one or two products (same M and r; separate partial buffers) in one launch + one reduce launch
int launch_wgrad(int precision, const WgradArgs* a, int n, hipStream_t s, ReduceQueue* defer) {
    // tokens per workgroup: at least WG_CHUNK, and large enough that the (channel blocks x chunks) grid of ONE product is
    // one round of the 256 CUs (B=128: 6 x 50 = 300 workgroups would leave a 44-workgroup second round; 6 x 40 does not);
    // a pair is two workgroups per CU, which hides the staging latency of the single-product launch
    const int M = a[0].M, r = a[0].r;
    if (dbg_skip(4)) return 0;
    if (n < 1 || n > 2 || (n == 2 && (a[1].M != M || a[1].r != r || a[1].partial == a[0].partial))) {
        set_error("launch_wgrad: bad pair");
        return -1;
    }
    const int cblocks = D / 128;
    const int want = (M + 256 / cblocks - 1) / (256 / cblocks);
    const int chunk = max(WG_CHUNK, (want + 63) / 64 * 64);
    const int nchunks = (M + chunk - 1) / chunk;
    WgPair src;
    WgOutPair outs;
    for (int i = 0; i < 2; ++i) {
        const WgradArgs& w = a[i < n ? i : 0];
        const bool half = precision == 0 && w.half_products;
        const float xs = half ? w.x_scale : 1.f, ys = half ? w.y_scale : 1.f;
        src.p[i] = WgSrc{w.X, w.Y, w.partial, xs, ys};
        outs.p[i] = WgOut{w.partial, w.out_w, w.sc, w.sj, w.alpha / (xs * ys), w.out_xsum, w.alpha_x / xs, w.out_ysum, w.alpha_y / ys};
    }
    if (precision == 0 && a[0].half_products != (n == 2 ? a[1].half_products : a[0].half_products)) { set_error("launch_wgrad: mixed product forms in a pair"); return -1; }
    if (precision == 0 && a[0].half_products) hipLaunchKernelGGL(wgrad_bf16_kernel<float>, dim3(D / 128, nchunks, n), dim3(256), 0, s, src, M, chunk);
    else if (precision == 0) hipLaunchKernelGGL(wgrad_f32_kernel, dim3(D / 128, nchunks, n), dim3(256), 0, s, src, M, chunk);
    else {
        // measurement hook: DYT_DBG_WGRAD_LDS = extra dynamic LDS bytes per workgroup (keeps other workgroups off the CU)
        static int extra = -1;
        if (extra < 0) {
            const char* e = getenv("DYT_DBG_WGRAD_LDS");
            extra = e ? atoi(e) : 0;
            if (extra > 0) DYT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, extra));
        }
        hipLaunchKernelGGL(wgrad_bf16_kernel<bf16>, dim3(D / 128, nchunks, n), dim3(256), extra, s, src, M, chunk);
    }
    if (defer) {
        if (defer->n_wg + n > ReduceQueue::MAX_WG || (defer->n_wg > 0 && defer->r != r)) { set_error("reduce queue full / mixed ranks"); return -1; }
        defer->r = r;
        for (int i = 0; i < n; ++i) {
            const WgOut& o = outs.p[i];
            defer->wg[defer->n_wg++] = WgReduceDesc{o.partial, o.out_w, o.sc, o.sj, o.alpha, o.out_xsum, o.alpha_x, o.out_ysum, o.alpha_y, nchunks};
        }
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(((D + 1) * (r + 1) + 255) / 256, n), dim3(256), 0, s, outs, nchunks, r);
    LAUNCH_CHECK();
    return 0;
}
int launch_wgrad(int precision, const WgradArgs& a, hipStream_t s) { return launch_wgrad(precision, &a, 1, s, nullptr); }

}  // namespace dyt
