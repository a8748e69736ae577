// Attentive pooling head of the video model (gfx950): final LayerNorm on ALL tokens, norm_q/k/v,
// q/k/v projections (q and v carry a bias, k does not), ONE query against the t*197 tokens of a clip
// per head, proj, classifier -- forward and backward.  Everything here is trainable (none of it is in
// the ViT checkpoint), so besides the dgrads this file produces the 14 pooling-head weight gradients.
//
// Reference ops replaced (paths relative to the reference root):
//   AttentiveBlock.forward / CrossAttention.forward   video_models/video_vision_transformer_IN21K.py:27-110
//   VisionTransformer.forward (pooling tail)          video_models/video_vision_transformer_IN21K.py:463-483
//
// The dense parts (K/V projections over b*t*197 rows, their dgrads and the two 768x768 weight
// gradients) go through the GEMM family of gemm.hip; the kernels below are the row-wise / tiny parts.
#include "kernels.h"
#include "rowhelp.h"

namespace dyt {

#define LAUNCH_CHECK() DYT_HIP_CHECK(hipGetLastError())

// ------------------------------------------------------------------------------------------
// xf = LN_final(x) (fp32, kept: it is the input of norm_k / norm_v) ; xk = LN_k(xf) ; xv = LN_v(xf)
// norm_k and norm_v see the same input, so they share (mean, rstd).
// ------------------------------------------------------------------------------------------
template <class AT>
__global__ __launch_bounds__(256) void pool_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ nw,
                                                          const float* __restrict__ nb, const float* __restrict__ kw,
                                                          const float* __restrict__ kb, const float* __restrict__ vw,
                                                          const float* __restrict__ vb, float* __restrict__ xf,
                                                          float2* __restrict__ st_f, float2* __restrict__ st_kv,
                                                          AT* __restrict__ xk, AT* __restrict__ xv, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    Row12 xr, w, b;
    xr.load(x + (size_t)row * D, lane);
    w.load(nw, lane);
    b.load(nb, lane);
    xr.landed(); w.landed(); b.landed();   // one full drain per load group (DYT_PIN*, dyt_common.h)
    const float2 s1 = ln_stats(xr);
#pragma unroll
    for (int i = 0; i < 12; ++i) xr.v[i] = (xr.v[i] - s1.x) * s1.y * w.v[i] + b.v[i];
    xr.store(xf + (size_t)row * D, lane);
    const float2 s2 = ln_stats(xr);
    if (lane == 0) { st_f[row] = s1; st_kv[row] = s2; }
    Row12 o;
    w.load(kw, lane);
    b.load(kb, lane);
    w.landed(); b.landed();
#pragma unroll
    for (int i = 0; i < 12; ++i) o.v[i] = (xr.v[i] - s2.x) * s2.y * w.v[i] + b.v[i];
    o.store(xk + (size_t)row * D, lane);
    w.load(vw, lane);
    b.load(vb, lane);
    w.landed(); b.landed();
#pragma unroll
    for (int i = 0; i < 12; ++i) o.v[i] = (xr.v[i] - s2.x) * s2.y * w.v[i] + b.v[i];
    o.store(xv + (size_t)row * D, lane);
}
int launch_pool_ln_fwd(int precision, const float* x, const float* nw, const float* nb, const float* kw, const float* kb,
                       const float* vw, const float* vb, float* xf, float2* st_f, float2* st_kv, void* xk, void* xv,
                       int rows, hipStream_t s) {
    const int grid = (rows + 3) / 4;
    if (precision == 0)
        hipLaunchKernelGGL(pool_ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, x, nw, nb, kw, kb, vw, vb, xf, st_f, st_kv,
                           (float*)xk, (float*)xv, rows);
    else
        hipLaunchKernelGGL(pool_ln_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, x, nw, nb, kw, kb, vw, vb, xf, st_f, st_kv,
                           (bf16*)xk, (bf16*)xv, rows);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// query path (batch independent): qn = LN_q(query_token) ; q = (Wq qn + q_bias) * head_dim^-0.5
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_q_ln_kernel(const float* __restrict__ query, const float* __restrict__ nqw,
                                                        const float* __restrict__ nqb, float* __restrict__ qn,
                                                        float* __restrict__ qhat, float* __restrict__ st_q) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float v[3], s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { v[i] = query[tid + 256 * i]; s += v[i]; }
    const float mean = block_sum256(s, red) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(block_sum256(q, red) * (1.0f / D) + LN_EPS);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid + 256 * i;
        const float xh = (v[i] - mean) * rstd;
        qn[c] = xh * nqw[c] + nqb[c];
        qhat[c] = xh;
    }
    if (tid == 0) { st_q[0] = mean; st_q[1] = rstd; }
}
int launch_rows_linear(const float* x, const float* W, const float* bias, float* out, int R, int N, int K, float scale,
                       hipStream_t s);
int launch_pool_q_fwd(const float* query, const float* nqw, const float* nqb, const float* Wq, const float* qbias, float* qn,
                      float* qhat, float* st_q, float* qs, hipStream_t s) {
    hipLaunchKernelGGL(pool_q_ln_kernel, dim3(1), dim3(256), 0, s, query, nqw, nqb, qn, qhat, st_q);
    LAUNCH_CHECK();
    return launch_rows_linear(qn, Wq, qbias, qs, 1, D, D, 0.125f, s);   // head_dim ** -0.5
}

// ------------------------------------------------------------------------------------------
// one query x NK keys per (clip, head): p = softmax(q.K^T), o = p V.  One workgroup per (clip, head);
// a wave owns a key (lane = head channel), fixed summation orders (deterministic).
// K, V: [clips*NK, 768] AT row-major (head h = columns h*64 .. h*64+63)
// ------------------------------------------------------------------------------------------
template <class AT>
__global__ __launch_bounds__(256) void pool_attn_fwd_kernel(const float* __restrict__ qs, const AT* __restrict__ K,
                                                            const AT* __restrict__ V, float* __restrict__ P,
                                                            float* __restrict__ o, int NK) {
    extern __shared__ float sm[];   // p[NK]
    __shared__ float red[4];
    __shared__ float ored[4][64];
    const int c = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const AT* Kc = K + (size_t)c * NK * D + h * HD;
    const AT* Vc = V + (size_t)c * NK * D + h * HD;
    const float qd = qs[h * HD + lane];
    for (int j = wave; j < NK; j += 4) {
        const float sc = wave_sum(qd * to_f32(Kc[(size_t)j * D + lane]));
        if (lane == 0) sm[j] = sc;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = tid; j < NK; j += 256) m = fmaxf(m, sm[j]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < NK; j += 256) { const float e = expf(sm[j] - m); sm[j] = e; sum += e; }
    sum = block_sum256(sum, red);
    const float inv = 1.0f / sum;
    float* Pc = P + ((size_t)c * NH + h) * NK;
    for (int j = tid; j < NK; j += 256) { const float p = sm[j] * inv; sm[j] = p; Pc[j] = p; }
    __syncthreads();
    float acc = 0.f;
    for (int j = wave; j < NK; j += 4) acc = fmaf(sm[j], to_f32(Vc[(size_t)j * D + lane]), acc);
    ored[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) o[(size_t)c * D + h * HD + lane] = ored[0][lane] + ored[1][lane] + ored[2][lane] + ored[3][lane];
}
int launch_pool_attn_fwd(int precision, const float* qs, const void* K, const void* V, float* P, float* o, int clips,
                         int NK, hipStream_t s) {
    const size_t sm = (size_t)NK * sizeof(float);
    if (sm > 60 * 1024) { set_error("pool_attn: %d keys per clip exceed the LDS budget", NK); return -1; }
    if (precision == 0)
        hipLaunchKernelGGL(pool_attn_fwd_kernel<float>, dim3(clips, NH), dim3(256), sm, s, qs, (const float*)K,
                           (const float*)V, P, o, NK);
    else
        hipLaunchKernelGGL(pool_attn_fwd_kernel<bf16>, dim3(clips, NH), dim3(256), sm, s, qs, (const bf16*)K,
                           (const bf16*)V, P, o, NK);
    LAUNCH_CHECK();
    return 0;
}

// backward of the above: dV_j = p_j do ; ds_j = p_j (do.V_j - sum_i p_i do.V_i) ; dK_j = ds_j q ;
// dq_part[clip] = sum_j ds_j K_j   (summed over clips later, in clip order)
template <class AT>
__global__ __launch_bounds__(256) void pool_attn_bwd_kernel(const float* __restrict__ qs, const AT* __restrict__ K,
                                                            const AT* __restrict__ V, const float* __restrict__ P,
                                                            const float* __restrict__ dO, AT* __restrict__ dK,
                                                            AT* __restrict__ dV, float* __restrict__ dq_part, int NK, float gs) {
    extern __shared__ float sm[];   // p[NK], dp[NK]
    __shared__ float red[4];
    __shared__ float qred[4][64];
    float* p = sm;
    float* dp = sm + NK;
    const int c = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)c * NK * D + h * HD;
    const float* Pc = P + ((size_t)c * NH + h) * NK;
    const float qd = qs[h * HD + lane];
    const float dod = dO[(size_t)c * D + h * HD + lane];
    for (int j = tid; j < NK; j += 256) p[j] = Pc[j];
    for (int j = wave; j < NK; j += 4) {
        const float t = wave_sum(dod * to_f32(V[base + (size_t)j * D + lane]));
        if (lane == 0) dp[j] = t;
    }
    __syncthreads();
    float spd = 0.f;
    for (int j = tid; j < NK; j += 256) spd = fmaf(p[j], dp[j], spd);
    spd = block_sum256(spd, red);
    float dq = 0.f;
    for (int j = wave; j < NK; j += 4) {
        const float pj = p[j];
        const float ds = pj * (dp[j] - spd);
        const size_t o = base + (size_t)j * D + lane;
        dq = fmaf(ds, to_f32(K[o]), dq);
        dK[o] = from_f32<AT>(ds * qd * gs);     // 16-bit gradient operands carry gs (fp16 build; 1 otherwise: exact)
        dV[o] = from_f32<AT>(pj * dod * gs);
    }
    qred[wave][lane] = dq;
    __syncthreads();
    if (wave == 0) dq_part[(size_t)c * D + h * HD + lane] = qred[0][lane] + qred[1][lane] + qred[2][lane] + qred[3][lane];
}
int launch_pool_attn_bwd(int precision, const float* qs, const void* K, const void* V, const float* P, const float* dO,
                         void* dK, void* dV, float* dq_part, int clips, int NK, float gs, hipStream_t s) {
    const size_t sm = (size_t)2 * NK * sizeof(float);
    if (sm > 60 * 1024) { set_error("pool_attn: %d keys per clip exceed the LDS budget", NK); return -1; }
    if (precision == 0)
        hipLaunchKernelGGL(pool_attn_bwd_kernel<float>, dim3(clips, NH), dim3(256), sm, s, qs, (const float*)K,
                           (const float*)V, P, dO, (float*)dK, (float*)dV, dq_part, NK, 1.0f);
    else
        hipLaunchKernelGGL(pool_attn_bwd_kernel<bf16>, dim3(clips, NH), dim3(256), sm, s, qs, (const bf16*)K,
                           (const bf16*)V, P, dO, (bf16*)dK, (bf16*)dV, dq_part, NK, gs);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// query path backward.  dq (w.r.t. the scaled q) = sum over clips of dq_part.
//   kernel W (grid 768): dWq[n,:] += g_n qn ; dq_bias[n] += g_n       with g_n = dq[n] / 8
//   then dqn = Wq^T g (rows_linear_bwd_x) and one workgroup for d norm_q.{w,b} ; d query_token
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_q_bwd_w_kernel(const float* __restrict__ dq_part, int clips,
                                                           const float* __restrict__ qn, float* __restrict__ dWq,
                                                           float* __restrict__ dqb, float* __restrict__ gq) {
    const int n = blockIdx.x, tid = threadIdx.x;
    float g = 0.f;
    for (int c = 0; c < clips; ++c) g += dq_part[(size_t)c * D + n];
    g *= 0.125f;
#pragma unroll
    for (int i = 0; i < 3; ++i) dWq[(size_t)n * D + tid + 256 * i] += g * qn[tid + 256 * i];
    if (tid == 0) { dqb[n] += g; gq[n] = g; }
}
__global__ void rows_linear_bwd_x_kernel(const float* __restrict__ dout, const float* __restrict__ W, float* __restrict__ dx,
                                         int R, int N, int K);   // defined with the tiny dense layers below
__global__ __launch_bounds__(256) void pool_q_bwd_ln_kernel(const float* __restrict__ dqn, const float* __restrict__ nqw,
                                                            const float* __restrict__ qhat, const float* __restrict__ st_q,
                                                            float* __restrict__ dnqw, float* __restrict__ dnqb,
                                                            float* __restrict__ dquery) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float dy[3], xh[3], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = tid + 256 * i;
        const float acc = dqn[k];
        xh[i] = qhat[k];
        dnqw[k] += acc * xh[i];
        dnqb[k] += acc;
        dy[i] = acc * nqw[k];
        s1 += dy[i];
        s2 = fmaf(dy[i], xh[i], s2);
    }
    s1 = block_sum256(s1, red) * (1.0f / D);
    s2 = block_sum256(s2, red) * (1.0f / D);
    const float rstd = st_q[1];
#pragma unroll
    for (int i = 0; i < 3; ++i) dquery[tid + 256 * i] += rstd * (dy[i] - s1 - xh[i] * s2);
}
int launch_pool_q_bwd(const float* dq_part, int clips, const float* qn, const float* qhat, const float* st_q, const float* Wq,
                      const float* nqw, float* gq, float* dqn, float* dWq, float* dqb, float* dnqw, float* dnqb,
                      float* dquery, hipStream_t s) {
    hipLaunchKernelGGL(pool_q_bwd_w_kernel, dim3(D), dim3(256), 0, s, dq_part, clips, qn, dWq, dqb, gq);
    hipLaunchKernelGGL(rows_linear_bwd_x_kernel, dim3(D / 64, 1), dim3(256), 0, s, gq, Wq, dqn, 1, D, D);   // dqn = Wq^T g
    hipLaunchKernelGGL(pool_q_bwd_ln_kernel, dim3(1), dim3(256), 0, s, dqn, nqw, qhat, st_q, dnqw, dnqb, dquery);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// tiny dense layers on R (= clips) rows: proj and the classifier, fp32
// ------------------------------------------------------------------------------------------
// out[r,n] = x[r,:] . W[n,:] + bias[n]        grid (ceil(N/4), R), one wave per output
__global__ __launch_bounds__(256) void rows_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ out, int N,
                                                          int K, float scale) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), r = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(x[(size_t)r * K + k], W[(size_t)n * K + k], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[(size_t)r * N + n] = (acc + (bias ? bias[n] : 0.f)) * scale;
}
int launch_rows_linear(const float* x, const float* W, const float* bias, float* out, int R, int N, int K, float scale,
                       hipStream_t s) {
    hipLaunchKernelGGL(rows_linear_kernel, dim3((N + 3) / 4, R), dim3(256), 0, s, x, W, bias, out, N, K, scale);
    LAUNCH_CHECK();
    return 0;
}
// dx[r,k] = sum_n dout[r,n] W[n,k].  Workgroup = 64 columns k (lane) x up to 16 rows r; the N loop is split over the
// 4 waves (fixed order), W is read once per workgroup, fully coalesced.   grid (K/64, ceil(R/16))
__global__ __launch_bounds__(256) void rows_linear_bwd_x_kernel(const float* __restrict__ dout, const float* __restrict__ W,
                                                                float* __restrict__ dx, int R, int N, int K) {
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane, r0 = blockIdx.y * 16;
    const int nr = min(16, R - r0);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int n = wave; n < N; n += 4) {
        const float w = W[(size_t)n * K + k];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < nr) acc[i] = fmaf(dout[(size_t)(r0 + i) * N + n], w, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    if (wave == 0)
        for (int i = 0; i < nr; ++i)
            dx[(size_t)(r0 + i) * K + k] = red[0][i][lane] + red[1][i][lane] + red[2][i][lane] + red[3][i][lane];
}
// dW[n,k] += sum_r dout[r,n] x[r,k] ; db[n] += sum_r dout[r,n]     grid N
__global__ __launch_bounds__(256) void rows_linear_bwd_w_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                                float* __restrict__ dW, float* __restrict__ db, int R, int N,
                                                                int K) {
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < K; k += 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(dout[(size_t)r * N + n], x[(size_t)r * K + k], acc);
        dW[(size_t)n * K + k] += acc;
    }
    if (tid == 0) {
        float sb = 0.f;
        for (int r = 0; r < R; ++r) sb += dout[(size_t)r * N + n];
        db[n] += sb;
    }
}
int launch_rows_linear_bwd(const float* dout, const float* x, const float* W, float* dx, float* dW, float* db, int R, int N,
                           int K, hipStream_t s) {
    if (K % 64 != 0) { set_error("rows_linear_bwd: K=%d must be a multiple of 64", K); return -1; }
    if (dx) hipLaunchKernelGGL(rows_linear_bwd_x_kernel, dim3(K / 64, (R + 15) / 16), dim3(256), 0, s, dout, W, dx, R, N, K);
    hipLaunchKernelGGL(rows_linear_bwd_w_kernel, dim3(N), dim3(256), 0, s, dout, x, dW, db, R, N, K);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// [rows, 768] AT -> [768, rows_pad] AT (zero padded), the K-contiguous operand layout of the NT GEMM that
// forms a 768x768 weight gradient (contraction over the token rows).  Optionally emits per-tile column
// sums (-> bias gradient): colsum_part[tile][768].
// ------------------------------------------------------------------------------------------
template <class AT>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const AT* __restrict__ src, AT* __restrict__ dst, int rows,
                                                             int rows_pad, float* __restrict__ colsum_part) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i;
        tile[i][tx] = r < rows ? to_f32(src[(size_t)r * D + c0 + tx]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) dst[(size_t)(c0 + i) * rows_pad + r0 + tx] = from_f32<AT>(tile[tx][i]);
    if (colsum_part && ty == 0) {
        float sacc = 0.f;
        for (int i = 0; i < 64; ++i) sacc += tile[i][tx];
        colsum_part[(size_t)blockIdx.x * D + c0 + tx] = sacc;
    }
}
int launch_transpose_rows(int precision, const void* src, void* dst, int rows, int rows_pad, float* colsum_part,
                          hipStream_t s) {
    const dim3 grid(rows_pad / 64, D / 64);
    if (precision == 0)
        hipLaunchKernelGGL(transpose_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, rows, rows_pad,
                           colsum_part);
    else
        hipLaunchKernelGGL(transpose_rows_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src, (bf16*)dst, rows, rows_pad,
                           colsum_part);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// backward through norm_k, norm_v and the final norm, one row per wave:
//   dxf = LNbwd_kv(dxk*gamma_k + dxv*gamma_v)  (norm_k / norm_v share xhat)  ;  g = LNbwd_final(dxf)
//   partial[blk] = { sum dxk*xhat, sum dxk, sum dxv*xhat, sum dxv }  (4 x 768, rows of this workgroup)
// ------------------------------------------------------------------------------------------
constexpr int POOL_ROWS_PER_BLOCK = 32;
template <class AT>
__global__ __launch_bounds__(256) void pool_ln_bwd_kernel(const AT* __restrict__ dxk, const AT* __restrict__ dxv,
                                                          const float* __restrict__ xf, const float2* __restrict__ st_kv,
                                                          const float* __restrict__ kw, const float* __restrict__ vw,
                                                          const float* __restrict__ x, const float2* __restrict__ st_f,
                                                          const float* __restrict__ nw, float* __restrict__ g,
                                                          float* __restrict__ partial, int rows, float inv_gs) {
    __shared__ float red[4][4 * D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Row12 gk, gv, wf, a_kx, a_k, a_vx, a_v;
    gk.load(kw, lane);
    gv.load(vw, lane);
    wf.load(nw, lane);
    gk.landed(); gv.landed(); wf.landed();
#pragma unroll
    for (int i = 0; i < 12; ++i) a_kx.v[i] = a_k.v[i] = a_vx.v[i] = a_v.v[i] = 0.f;
    const int r0 = blockIdx.x * POOL_ROWS_PER_BLOCK;
    for (int k = wave; k < POOL_ROWS_PER_BLOCK; k += 4) {
        const int row = r0 + k;
        if (row >= rows) break;
        Row12 dk, dv, xr;
        dk.load_at(dxk + (size_t)row * D, lane);
        dv.load_at(dxv + (size_t)row * D, lane);
        xr.load(xf + (size_t)row * D, lane);
        float2 s2 = st_kv[row];
        float2 sf = st_f[row];
        Row12 x0;
        x0.load(x + (size_t)row * D, lane);
        dk.landed(); dv.landed(); xr.landed(); x0.landed();
#pragma unroll
        for (int i = 0; i < 12; ++i) { dk.v[i] *= inv_gs; dv.v[i] *= inv_gs; }   // the 16-bit operands carried gs
        DYT_PIN4(s2.x, s2.y, sf.x, sf.y);   // the per-row statistics: the loads whose early consumption made ln_bwd irreproducible
        Row12 dy;
        float s1 = 0.f, sx = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float xh = (xr.v[i] - s2.x) * s2.y;
            a_kx.v[i] = fmaf(dk.v[i], xh, a_kx.v[i]); a_k.v[i] += dk.v[i];
            a_vx.v[i] = fmaf(dv.v[i], xh, a_vx.v[i]); a_v.v[i] += dv.v[i];
            dy.v[i] = dk.v[i] * gk.v[i] + dv.v[i] * gv.v[i];
            s1 += dy.v[i];
            sx = fmaf(dy.v[i], xh, sx);
            xr.v[i] = xh;
        }
        s1 = wave_sum(s1) * (1.0f / D);
        sx = wave_sum(sx) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < 12; ++i) dy.v[i] = s2.y * (dy.v[i] - s1 - xr.v[i] * sx);   // dL/dxf
        ln_bwd_row(dy, x0, wf, sf);
        dy.store(g + (size_t)row * D, lane);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = i * 256 + lane * 4 + e;
            red[wave][c] = a_kx.v[4 * i + e];
            red[wave][D + c] = a_k.v[4 * i + e];
            red[wave][2 * D + c] = a_vx.v[4 * i + e];
            red[wave][3 * D + c] = a_v.v[4 * i + e];
        }
    __syncthreads();
    for (int c = tid; c < 4 * D; c += 256)
        partial[(size_t)blockIdx.x * 4 * D + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}
int launch_pool_ln_bwd(int precision, const void* dxk, const void* dxv, const float* xf, const float2* st_kv, const float* kw,
                       const float* vw, const float* x, const float2* st_f, const float* nw, float* g, float* partial,
                       int rows, int* nblocks_out, float gs, hipStream_t s) {
    const int grid = (rows + POOL_ROWS_PER_BLOCK - 1) / POOL_ROWS_PER_BLOCK;
    if (nblocks_out) *nblocks_out = grid;
    if (precision == 0)
        hipLaunchKernelGGL(pool_ln_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dxk, (const float*)dxv, xf,
                           st_kv, kw, vw, x, st_f, nw, g, partial, rows, 1.0f);
    else
        hipLaunchKernelGGL(pool_ln_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)dxk, (const bf16*)dxv, xf,
                           st_kv, kw, vw, x, st_f, nw, g, partial, rows, 1.0f / gs);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace dyt
