// Multi-head self-attention for the DyT ViT-B/16 path: N = 197 tokens, 12 heads x 64.
// Replaces Attention.forward's F.scaled_dot_product_attention / explicit softmax
// (models/vision_transformer_IN21K.py:60-70 of the reference) and its autograd backward.
//
// N = 197 is small: K and V of one (image, head) are 25 KB in bf16 and live in LDS for the
// whole workgroup; a wave keeps the complete 32 x 224 score block of its query tile in
// registers, so no online-softmax rescaling is needed (197 is padded to 224 = 7 x 32 and the
// pad keys are masked to -inf).
//
// fast path (bf16, v_mfma_f32_32x32x16_bf16).  Layout tricks, all addressing-only:
//   * scores are computed TRANSPOSED, S^T = K Q^T, so each lane owns one query column: the row
//     max / row sum are in-lane reductions plus a single cross-half shuffle;
//   * the MFMA C/D layout of that tile (lane <-> query, registers <-> keys) is already the
//     B-operand layout of the next MFMA (O^T = V^T P^T) up to a fixed permutation of the k
//     index; the A operand (V^T from an LDS-transposed copy) is fetched with the SAME
//     permutation, so P never moves between lanes;
//   * the same holds for every product of the backward pass (two kernels: dQ per query tile,
//     dK/dV per key tile -- no atomics, deterministic).
// exact path (fp32 operands on the matrix cores, v_mfma_f32_32x32x2_f32): the parity mode, same tiling.
#include "kernels.h"

namespace dyt {

constexpr int NPAD = 224;   // 7 tiles of 32
constexpr int RLD = 72;     // bf16 per row of a row-major [224][64] LDS image (144 B: conflict-free b128 fragment reads)
constexpr int TLD = 228;    // bf16 per row of a transposed [64][224] LDS image (456 B = 8 x odd: conflict-free b64 reads)
constexpr int ROW_IMG = NPAD * RLD * 2;  // 32256 B
constexpr int TR_IMG = HD * TLD * 2;     // 29184 B

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int base) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)v[base + i];
    return o;
}
__device__ __forceinline__ bf16x8 join44(const bf16* p) {  // p[0..3] and p[8..11]
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(p + 8);
    bf16x8 o = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return o;
}

// stage a [197][64] matrix (row stride `ld` elements in global) into a row-major and/or a
// transposed LDS image; rows >= 197 are zero.
template <int NTHREADS>
__device__ __forceinline__ void stage_rows(const bf16* __restrict__ src, int ld, bf16* rowimg, bf16* trimg, int tid) {
    for (int t = tid; t < (NPAD / 2) * 8; t += NTHREADS) {
        const int pr = t >> 3, c = t & 7, r0 = pr * 2;
        bf16x8 a = zero8(), b = zero8();
        if (r0 < NT) a = *reinterpret_cast<const bf16x8*>(src + (size_t)r0 * ld + c * 8);
        if (r0 + 1 < NT) b = *reinterpret_cast<const bf16x8*>(src + (size_t)(r0 + 1) * ld + c * 8);
        if (rowimg) {
            *reinterpret_cast<bf16x8*>(rowimg + r0 * RLD + c * 8) = a;
            *reinterpret_cast<bf16x8*>(rowimg + (r0 + 1) * RLD + c * 8) = b;
        }
        if (trimg) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x2 pv = {a[i], b[i]};
                *reinterpret_cast<bf16x2*>(trimg + (c * 8 + i) * TLD + r0) = pv;
            }
        }
    }
}

// The same staging in two halves, so that a persistent workgroup can have the NEXT head's rows in flight (registers) while it
// computes on the current one: 896 tasks (row pair x 16-B chunk) over 448 threads = 2 tasks = 4 x 16 B per thread and matrix.
struct StageRegs { bf16x8 a[2], b[2]; };
__device__ __forceinline__ void stage_load(StageRegs& r, const bf16* __restrict__ src, int ld, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + u * 448, pr = t >> 3, c = t & 7, r0 = pr * 2;
        // unconditional loads from clamped rows (no exec-masked branches in front of the loads); pad rows are zeroed afterwards
        r.a[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)min(r0, NT - 1) * ld + c * 8);
        r.b[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)min(r0 + 1, NT - 1) * ld + c * 8);
    }
}
__device__ __forceinline__ void stage_store(const StageRegs& r, bf16* rowimg, bf16* trimg, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + u * 448, pr = t >> 3, c = t & 7, r0 = pr * 2;
        const bf16x8 a = r0 < NT ? r.a[u] : zero8(), b = r0 + 1 < NT ? r.b[u] : zero8();
        if (rowimg) {
            *reinterpret_cast<bf16x8*>(rowimg + r0 * RLD + c * 8) = a;
            *reinterpret_cast<bf16x8*>(rowimg + (r0 + 1) * RLD + c * 8) = b;
        }
        if (trimg) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x2 pv = {a[i], b[i]};
                *reinterpret_cast<bf16x2*>(trimg + (c * 8 + i) * TLD + r0) = pv;
            }
        }
    }
}

#define MFMA32(a, b, c) DYT_MFMA_32x32x16((a), (b), (c))

// ------------------------------------------------------------------------------------------
// forward, bf16
// ------------------------------------------------------------------------------------------
// Persistent: workgroup w walks the (image, head) pairs w, w + gridDim.x, ...; while it computes on one head the K, V and
// Q rows of the next are already in flight into registers (the staging's HBM/L2 latency was 1/3 of the kernel).
__global__ __launch_bounds__(448) void attn_fwd_bf16_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                            const bf16* __restrict__ v, bf16* __restrict__ out,
                                                            float* __restrict__ lse, int nheads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Ks = reinterpret_cast<bf16*>(smem);
    bf16* Vt = reinterpret_cast<bf16*>(smem + ROW_IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31, qr = min(qrow, NT - 1);   // one 32-row query tile per wave (7 waves)
    StageRegs kr, vr;
    bf16x8 qn[4];
    auto prefetch = [&](int bh) {
        stage_load(kr, k + (size_t)bh * NT * HD, HD, tid);
        stage_load(vr, v + (size_t)bh * NT * HD, HD, tid);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qn[ks] = *reinterpret_cast<const bf16x8*>(q + ((size_t)bh * NT + qr) * HD + ks * 16 + hi * 8);
    };
    int bh = blockIdx.x;
    if (bh < nheads) prefetch(bh);
    for (; bh < nheads; bh += gridDim.x) {
        const int b = bh / NH, h = bh - b * NH;
        stage_store(kr, Ks, nullptr, tid);
        stage_store(vr, nullptr, Vt, tid);
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
        __syncthreads();
        if (bh + gridDim.x < nheads) prefetch(bh + gridDim.x);   // lands while this head is computed
        f32x16 st[7];
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
        // S^T = K Q^T: the seven key tiles are independent accumulators -- walk them inside each k step so that consecutive
        // MFMAs never depend on each other, and read a whole k step's fragments ahead of its MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[7];
#pragma unroll
            for (int kt = 0; kt < 7; ++kt) a[kt] = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + l31) * RLD + ks * 16 + hi * 8);
#pragma unroll
            for (int kt = 0; kt < 7; ++kt) st[kt] = MFMA32(a[kt], qf[ks], st[kt]);
        }
        // st[kt][r] = S^T[key = kt*32 + (r&3) + 8*(r>>2) + 4*hi][q = l31]; mask the pad keys
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 192 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= NT) st[6][r] = -INFINITY;
        }
        float mp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // independent chains: the reductions are latency-, not issue-bound
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mp[r & 3] = fmaxf(mp[r & 3], st[kt][r]);
        float m = fmaxf(fmaxf(mp[0], mp[1]), fmaxf(mp[2], mp[3]));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        // exp, row sum and O^T = V^T P^T per key tile in one loop: the 4 MFMAs of tile kt execute while the VALU works on the
        // exponentials of tile kt+1 (as two separate loops the matrix pipe idled through the whole softmax and vice versa)
        float sp[4] = {0.f, 0.f, 0.f, 0.f};
        f32x16 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < 7; ++kt) {
            bf16x8 vf[2][2];
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) vf[half][dt] = join44(Vt + (dt * 32 + l31) * TLD + kt * 32 + half * 16 + 4 * hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(st[kt][r] - m);
                st[kt][r] = p;
                sp[r & 3] += p;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8 pf = pack8(st[kt], half * 8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = MFMA32(vf[half][dt], pf, o[dt]);
            }
        }
        float sum = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        sum += __shfl_xor(sum, 32, 64);
        if (hi == 0 && qrow < NT) lse[(size_t)bh * NT + qrow] = m + __logf(sum);
        if (qrow < NT) {
            const float inv = 1.0f / sum;
            bf16* op = out + ((size_t)b * NT + qrow) * D + h * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    store4(op + dt * 32 + 8 * g + 4 * hi, o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv,
                           o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        }
        __syncthreads();   // every wave is done with this head's K / V images before they are overwritten
    }
}

// ------------------------------------------------------------------------------------------
// backward, bf16: dQ (+ delta = rowsum(dO * O)) per query tile
// ------------------------------------------------------------------------------------------
// Persistent over heads like the forward kernel (next head's K, V and this wave's q / dO / O rows in flight during the
// current one).
__global__ __launch_bounds__(448) void attn_bwd_dq_bf16_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                               const bf16* __restrict__ v, const bf16* __restrict__ o,
                                                               const bf16* __restrict__ dout,
                                                               const float* __restrict__ lse, float* __restrict__ delta,
                                                               bf16* __restrict__ dqkv, int nheads, int o_ld) {
    // o_ld: row stride of `o` in elements (768; 1536 when o is the hi plane of a [M][hi | ...] split operand image)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Ks = reinterpret_cast<bf16*>(smem);
    bf16* Vs = reinterpret_cast<bf16*>(smem + ROW_IMG);
    bf16* Kt = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31, qr = min(qrow, NT - 1);   // one 32-row query tile per wave (7 waves)
    StageRegs kr, vr;
    bf16x8 qn[4], don[4], on[4];
    float Ln = 0.f;
    auto prefetch = [&](int bh) {
        const int b = bh / NH, h = bh - b * NH;
        stage_load(kr, k + (size_t)bh * NT * HD, HD, tid);
        stage_load(vr, v + (size_t)bh * NT * HD, HD, tid);
        const size_t trow = ((size_t)b * NT + qr) * D + h * HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qn[ks] = *reinterpret_cast<const bf16x8*>(q + ((size_t)bh * NT + qr) * HD + ks * 16 + hi * 8);
            don[ks] = *reinterpret_cast<const bf16x8*>(dout + trow + ks * 16 + hi * 8);
            on[ks] = *reinterpret_cast<const bf16x8*>(o + ((size_t)b * NT + qr) * o_ld + h * HD + ks * 16 + hi * 8);
        }
        Ln = lse[(size_t)bh * NT + qr];
    };
    int bh = blockIdx.x;
    if (bh < nheads) prefetch(bh);
    for (; bh < nheads; bh += gridDim.x) {
        const int b = bh / NH, h = bh - b * NH;
        stage_store(kr, Ks, Kt, tid);
        stage_store(vr, Vs, nullptr, tid);
        bf16x8 qf[4], dof[4];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = qn[ks]; dof[ks] = don[ks];
#pragma unroll
            for (int i = 0; i < 8; ++i) dl += (float)don[ks][i] * (float)on[ks][i];
        }
        const float L = Ln;
        dl += __shfl_xor(dl, 32, 64);
        if (hi == 0 && qrow < NT) delta[(size_t)bh * NT + qrow] = dl;
        __syncthreads();
        if (bh + gridDim.x < nheads) prefetch(bh + gridDim.x);

        f32x16 dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
        auto scores = [&](int kt, f32x16& s_, f32x16& dp_) {   // S^T[key][q], dP^T[key][q] of key tile kt
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp_[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 ka = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + l31) * RLD + ks * 16 + hi * 8);
                const bf16x8 va = *reinterpret_cast<const bf16x8*>(Vs + (kt * 32 + l31) * RLD + ks * 16 + hi * 8);
                s_ = MFMA32(ka, qf[ks], s_);
                dp_ = MFMA32(va, dof[ks], dp_);
            }
        };
#pragma unroll 1
        for (int kt = 0; kt < 7; ++kt) {
            f32x16 s_, dp_;
            scores(kt, s_, dp_);
            bf16x8 kf[2][2];
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) kf[half][dt] = join44(Kt + (dt * 32 + l31) * TLD + kt * 32 + half * 16 + 4 * hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float p = key < NT ? __expf(s_[r] - L) : 0.f;
                s_[r] = p * (dp_[r] - dl);        // dS^T
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8 dsf = pack8(s_, half * 8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt] = MFMA32(kf[half][dt], dsf, dq[dt]);  // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            }
        }
        if (qrow < NT) {
            bf16* op = dqkv + ((size_t)b * NT + qrow) * (3 * D) + h * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    store4(op + dt * 32 + 8 * g + 4 * hi, dq[dt][4 * g] * 0.125f, dq[dt][4 * g + 1] * 0.125f,
                           dq[dt][4 * g + 2] * 0.125f, dq[dt][4 * g + 3] * 0.125f);
        }
        __syncthreads();   // every wave is done with this head's images before they are overwritten
    }
}

// ------------------------------------------------------------------------------------------
// backward, bf16: dK, dV per key tile
// ------------------------------------------------------------------------------------------
// 7 waves, one 32-key tile each; persistent over heads with the next head's Q / dO rows (and lse, delta) in flight.
__global__ __launch_bounds__(448) void attn_bwd_dkv_bf16_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                                const bf16* __restrict__ v,
                                                                const bf16* __restrict__ dout,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta,
                                                                bf16* __restrict__ dqkv, int nheads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Qs = reinterpret_cast<bf16*>(smem);
    bf16* dOs = reinterpret_cast<bf16*>(smem + ROW_IMG);
    bf16* Qt = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG);
    bf16* dOt = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG + TR_IMG);
    float* lse_s = reinterpret_cast<float*>(smem + 2 * ROW_IMG + 2 * TR_IMG);
    float* del_s = lse_s + NPAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int key = wave * 32 + l31, kr_ = min(key, NT - 1);
    StageRegs qr, dr;
    float ln = 0.f, dn = 0.f;
    auto prefetch = [&](int bh) {
        const int b = bh / NH, h = bh - b * NH;
        stage_load(qr, q + (size_t)bh * NT * HD, HD, tid);
        stage_load(dr, dout + (size_t)b * NT * D + h * HD, D, tid);
        if (tid < NPAD) {
            ln = tid < NT ? lse[(size_t)bh * NT + tid] : 0.f;
            dn = tid < NT ? delta[(size_t)bh * NT + tid] : 0.f;
        }
    };
    int bh = blockIdx.x;
    if (bh < nheads) prefetch(bh);
    for (; bh < nheads; bh += gridDim.x) {
        const int b = bh / NH, h = bh - b * NH;
        const bf16* kb = k + ((size_t)bh * NT + kr_) * HD;
        const bf16* vb = v + ((size_t)bh * NT + kr_) * HD;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {   // this head's key / value rows of the wave's tile: land during the staging stores
            kf[ks] = *reinterpret_cast<const bf16x8*>(kb + ks * 16 + hi * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vb + ks * 16 + hi * 8);
        }
        stage_store(qr, Qs, Qt, tid);
        stage_store(dr, dOs, dOt, tid);
        if (tid < NPAD) { lse_s[tid] = ln; del_s[tid] = dn; }
        __syncthreads();
        if (bh + gridDim.x < nheads) prefetch(bh + gridDim.x);
        f32x16 aK[2], aV[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { aK[0][r] = 0.f; aK[1][r] = 0.f; aV[0][r] = 0.f; aV[1][r] = 0.f; }

#pragma unroll 1
        for (int qt = 0; qt < 7; ++qt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + (qt * 32 + l31) * RLD + ks * 16 + hi * 8);
                const bf16x8 da = *reinterpret_cast<const bf16x8*>(dOs + (qt * 32 + l31) * RLD + ks * 16 + hi * 8);
                s = MFMA32(qa, kf[ks], s);     // S[q][key]   (rows q in registers, column key = lane)
                dp = MFMA32(da, vf[ks], dp);   // dP[q][key]
            }
            f32x16 p;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q0 = qt * 32 + 8 * g + 4 * hi;
                const float4 L4 = *reinterpret_cast<const float4*>(lse_s + q0);
                const float4 D4 = *reinterpret_cast<const float4*>(del_s + q0);
                const float Ls[4] = {L4.x, L4.y, L4.z, L4.w};
                const float Ds[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const bool ok = (q0 + e < NT) && (key < NT);
                    const float pv = ok ? __expf(s[r] - Ls[e]) : 0.f;
                    p[r] = pv;
                    s[r] = pv * (dp[r] - Ds[e]);  // dS
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x8 pf = pack8(p, half * 8);
                const bf16x8 dsf = pack8(s, half * 8);
                const int qbase = qt * 32 + half * 16 + 4 * hi;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 dot = join44(dOt + (dt * 32 + l31) * TLD + qbase);
                    const bf16x8 qtf = join44(Qt + (dt * 32 + l31) * TLD + qbase);
                    aV[dt] = MFMA32(dot, pf, aV[dt]);   // dV^T[d][key] += dO^T[d][q] P[q][key]
                    aK[dt] = MFMA32(qtf, dsf, aK[dt]);  // dK^T[d][key] += Q^T[d][q] dS[q][key]
                }
            }
        }
        if (key < NT) {
            bf16* op = dqkv + ((size_t)b * NT + key) * (3 * D) + h * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * hi;
                    store4(op + D + d, aK[dt][4 * g], aK[dt][4 * g + 1], aK[dt][4 * g + 2], aK[dt][4 * g + 3]);
                    store4(op + 2 * D + d, aV[dt][4 * g], aV[dt][4 * g + 1], aV[dt][4 * g + 2], aV[dt][4 * g + 3]);
                }
        }
        __syncthreads();   // every wave is done with this head's images before they are overwritten
    }
}

// ------------------------------------------------------------------------------------------
// backward, bf16: dQ and dK / dV of one (image, head) in ONE workgroup pass
// ------------------------------------------------------------------------------------------
// The two kernels above each stream q, k, v, dO (and o) of every head from memory: 12 passes over a [M,768] operand per
// backward (464 MB at B=128) at ~3 TB/s -- the attention backward is bound by that traffic and by per-head latencies, not by
// the matrix pipe.  Here a persistent workgroup runs the dK/dV phase and the dQ phase of a head back to back on the same LDS:
//   phase B  Q, dO staged once (row + transposed images); the wave's k / v rows in registers; delta = rowsum(dO * o) goes
//            from registers to LDS (no trip through memory) -> dK, dV   (the code of attn_bwd_dkv_bf16_kernel)
//   switch   every wave takes the q / dO rows of ITS query tile from the row images (phase A's B operands), then the K / V
//            images (row + K^T) replace Q / dO: their cooperative loads were issued before phase B and landed under it
//   phase A  dQ   (the code of attn_bwd_dq_bf16_kernel)
// Q, dO and o are read once, k / v twice by the same workgroup within one head (second read from L2): 8 operand passes
// instead of 12, one launch instead of two.  The heavy phase (B: four accumulators) runs with 64 prefetch registers in
// flight, the light one (A) with the next head's 112 -- in the opposite order the kernel spills.  The arithmetic of both
// phases is the code of the two kernels above, so the results are bit-identical to theirs (tested).
template <bool ROWPF>
__global__ __launch_bounds__(448) void attn_bwd_fused_bf16_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                                  const bf16* __restrict__ v, const bf16* __restrict__ o,
                                                                  const bf16* __restrict__ dout,
                                                                  const float* __restrict__ lse, float* __restrict__ delta,
                                                                  bf16* __restrict__ dqkv, int nheads, int nq, int o_ld) {
    // nq: query tiles (of 32 rows) that can carry a non-zero dO -- 7, or 1 when only the cls rows have an upstream gradient
    // (last block, cls-only tail): rows with dO = 0 have dP = delta = 0, hence dS = 0 and no contribution to dK / dV / dQ, so
    // phase B walks nq query tiles and only the first nq waves run phase A (the others store their zero dQ rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // phase B images
    bf16* Qs = reinterpret_cast<bf16*>(smem);
    bf16* dOs = reinterpret_cast<bf16*>(smem + ROW_IMG);
    bf16* Qt = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG);
    bf16* dOt = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG + TR_IMG);
    float* lse_s = reinterpret_cast<float*>(smem + 2 * ROW_IMG + 2 * TR_IMG);
    float* del_s = lse_s + NPAD;
    // phase A images (same bytes)
    bf16* Ks = reinterpret_cast<bf16*>(smem);
    bf16* Vs = reinterpret_cast<bf16*>(smem + ROW_IMG);
    bf16* Kt = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31, qr = min(qrow, NT - 1);   // the wave's key tile (phase B) = its query tile (phase A)
    StageRegs qs_, ds_;            // next head's Q / dO (cooperative): in flight during phase A
    bf16x8 kfr[4], vfr[4], on[4];  // k / v / o rows of this wave's tile
    float Ln = 0.f;
    auto load_rows = [&](int bh) {
        const int b = bh / NH, h = bh - b * NH;
        const size_t hrow = ((size_t)bh * NT + qr) * HD;
        const size_t trow = ((size_t)b * NT + qr) * D + h * HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kfr[ks] = *reinterpret_cast<const bf16x8*>(k + hrow + ks * 16 + hi * 8);
            vfr[ks] = *reinterpret_cast<const bf16x8*>(v + hrow + ks * 16 + hi * 8);
            on[ks] = *reinterpret_cast<const bf16x8*>(o + ((size_t)b * NT + qr) * o_ld + h * HD + ks * 16 + hi * 8);
        }
        Ln = lse[(size_t)bh * NT + qr];
    };
    auto prefetch = [&](int bh) {
        const int b = bh / NH, h = bh - b * NH;
        stage_load(qs_, q + (size_t)bh * NT * HD, HD, tid);
        stage_load(ds_, dout + (size_t)b * NT * D + h * HD, D, tid);
        if (ROWPF) load_rows(bh);   // else: issued at the head's start, first used after the image stores and two barriers
    };
    int bh = blockIdx.x;
    if (bh < nheads) prefetch(bh);
    for (; bh < nheads; bh += gridDim.x) {
        const int b = bh / NH, h = bh - b * NH;
        if (!ROWPF) load_rows(bh);
        stage_store(qs_, Qs, Qt, tid);
        stage_store(ds_, dOs, dOt, tid);
        const float L = Ln;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = kfr[ks]; vf[ks] = vfr[ks]; }
        __syncthreads();
        // delta of this wave's query rows: dO rows from the image just written (the operand order of the dQ kernel's sum)
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dOs + qrow * RLD + ks * 16 + hi * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) dl += (float)dv[i] * (float)on[ks][i];
        }
        dl += __shfl_xor(dl, 32, 64);
        if (hi == 0) {
            lse_s[qrow] = qrow < NT ? L : 0.f;
            del_s[qrow] = qrow < NT ? dl : 0.f;
            if (qrow < NT) delta[(size_t)bh * NT + qrow] = dl;
        }
        __syncthreads();
        StageRegs kr, vr;   // this head's K / V for the phase-A images: in flight during phase B
        stage_load(kr, k + (size_t)bh * NT * HD, HD, tid);
        stage_load(vr, v + (size_t)bh * NT * HD, HD, tid);
        // ---------------- phase B: dK, dV ----------------
        {
            const int key = qrow;
            f32x16 aK[2], aV[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { aK[0][r] = 0.f; aK[1][r] = 0.f; aV[0][r] = 0.f; aV[1][r] = 0.f; }
#pragma unroll 1
            for (int qt = 0; qt < nq; ++qt) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + (qt * 32 + l31) * RLD + ks * 16 + hi * 8);
                    const bf16x8 da = *reinterpret_cast<const bf16x8*>(dOs + (qt * 32 + l31) * RLD + ks * 16 + hi * 8);
                    s = MFMA32(qa, kf[ks], s);     // S[q][key]   (rows q in registers, column key = lane)
                    dp = MFMA32(da, vf[ks], dp);   // dP[q][key]
                }
                f32x16 p;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q0 = qt * 32 + 8 * g + 4 * hi;
                    const float4 L4 = *reinterpret_cast<const float4*>(lse_s + q0);
                    const float4 D4 = *reinterpret_cast<const float4*>(del_s + q0);
                    const float Ls[4] = {L4.x, L4.y, L4.z, L4.w};
                    const float Ds[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const bool ok = (q0 + e < NT) && (key < NT);
                        const float pv = ok ? __expf(s[r] - Ls[e]) : 0.f;
                        p[r] = pv;
                        s[r] = pv * (dp[r] - Ds[e]);  // dS
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const bf16x8 pf = pack8(p, half * 8);
                    const bf16x8 dsf = pack8(s, half * 8);
                    const int qbase = qt * 32 + half * 16 + 4 * hi;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const bf16x8 dot = join44(dOt + (dt * 32 + l31) * TLD + qbase);
                        const bf16x8 qtf = join44(Qt + (dt * 32 + l31) * TLD + qbase);
                        aV[dt] = MFMA32(dot, pf, aV[dt]);   // dV^T[d][key] += dO^T[d][q] P[q][key]
                        aK[dt] = MFMA32(qtf, dsf, aK[dt]);  // dK^T[d][key] += Q^T[d][q] dS[q][key]
                    }
                }
            }
            if (key < NT) {
                bf16* op = dqkv + ((size_t)b * NT + key) * (3 * D) + h * HD;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d = dt * 32 + 8 * g + 4 * hi;
                        store4(op + D + d, aK[dt][4 * g], aK[dt][4 * g + 1], aK[dt][4 * g + 2], aK[dt][4 * g + 3]);
                        store4(op + 2 * D + d, aV[dt][4 * g], aV[dt][4 * g + 1], aV[dt][4 * g + 2], aV[dt][4 * g + 3]);
                    }
            }
        }
        // ---------------- switch ----------------
        bf16x8 qf[4], dof[4];   // q / dO rows of this wave's query tile (pad rows are zero in the images; their columns are never stored)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(Qs + qrow * RLD + ks * 16 + hi * 8);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dOs + qrow * RLD + ks * 16 + hi * 8);
        }
        __syncthreads();   // every wave is done with the Q / dO images
        stage_store(kr, Ks, Kt, tid);
        stage_store(vr, Vs, nullptr, tid);
        __syncthreads();
        if (bh + gridDim.x < nheads) prefetch(bh + gridDim.x);   // lands under phase A
        // ---------------- phase A: dQ ----------------
        {
            f32x16 dq[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
#pragma unroll 1
            for (int kt = 0; kt < (wave < nq ? 7 : 0); ++kt) {
                f32x16 s_, dp_;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp_[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 ka = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + l31) * RLD + ks * 16 + hi * 8);
                    const bf16x8 va = *reinterpret_cast<const bf16x8*>(Vs + (kt * 32 + l31) * RLD + ks * 16 + hi * 8);
                    s_ = MFMA32(ka, qf[ks], s_);
                    dp_ = MFMA32(va, dof[ks], dp_);
                }
                bf16x8 ktf[2][2];
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) ktf[half][dt] = join44(Kt + (dt * 32 + l31) * TLD + kt * 32 + half * 16 + 4 * hi);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float p = key < NT ? __expf(s_[r] - L) : 0.f;
                    s_[r] = p * (dp_[r] - dl);        // dS^T
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const bf16x8 dsf = pack8(s_, half * 8);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) dq[dt] = MFMA32(ktf[half][dt], dsf, dq[dt]);  // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
                }
            }
            if (qrow < NT) {
                bf16* op = dqkv + ((size_t)b * NT + qrow) * (3 * D) + h * HD;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        store4(op + dt * 32 + 8 * g + 4 * hi, dq[dt][4 * g] * 0.125f, dq[dt][4 * g + 1] * 0.125f,
                               dq[dt][4 * g + 2] * 0.125f, dq[dt][4 * g + 3] * 0.125f);
            }
        }
        __syncthreads();   // every wave is done with this head's images before they are overwritten
    }
}

// ------------------------------------------------------------------------------------------
// exact fp32 kernels on the matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulate, one
// rounding per product = an fmaf chain) -- the parity mode.
//
// Same decomposition as the bf16 kernels (one 32-row query / key tile per wave, 7 waves, the full 32 x 224 score
// block of a tile in registers, scores transposed so the softmax is lane-local, C/D layout of one product = B
// operand of the next).  The operand of a 32x32x2 MFMA is ONE fp32 per lane (lane l: row l & 31, k = l >> 5), so:
//   * a row fragment (k = head channel d) is a ds_read_b128 of 4 consecutive d -- lanes 0..31 take d < 32, lanes
//     32..63 d >= 32, register t of the chunk feeds MFMA t; both operands use the same assignment, only the order in
//     which the 64 channels are accumulated is permuted.  Row stride 68 floats: conflict-free for that read;
//   * a TRANSPOSED fragment (k = key / query index, rows = d) is a ds_read_b32 down a column of the same row-major
//     image with the lanes along d -- no transposed LDS copies at all (the bf16 kernels need two).
// Pad rows (197..223) are zero-filled so that 0 * pad stays 0.
// ------------------------------------------------------------------------------------------
constexpr int FLD = 68;                      // fp32 per LDS row (272 B)
constexpr int F_IMG = NPAD * FLD * 4;        // 60928 B
#define MFMA32F(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// stage a [197][64] fp32 matrix (row stride `ld` in global) into a row-major LDS image with row stride LD; rows >= 197 zero
template <int NTHREADS, int LD>
__device__ __forceinline__ void stage_rows_f32(const float* __restrict__ src, int ld, float* img, int tid) {
    for (int t = tid; t < NPAD * 16; t += NTHREADS) {
        const int r = t >> 4, c = t & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < NT) v = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c * 4);
        *reinterpret_cast<float4*>(img + r * LD + c * 4) = v;
    }
}
// the 32 channels [hi*32, hi*32+32) of one global row: the B operand of a row-fragment product, register s <-> d = hi*32 + s
__device__ __forceinline__ void load_half_row(const float* __restrict__ p, float (&o)[32]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(p + c * 4);
        o[4 * c] = v.x; o[4 * c + 1] = v.y; o[4 * c + 2] = v.z; o[4 * c + 3] = v.w;
    }
}
// acc[rows of img tile][cols of breg] += sum_d img[row0 + (lane&31)][d] * breg[d]   (d permuted as described above)
__device__ __forceinline__ void mma_rowfrag(f32x16& acc, const float* img, int row0, int l31, int hi, const float (&breg)[32]) {
    const float* rp = img + (row0 + l31) * FLD + hi * 32;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(rp + c * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = MFMA32F(a[t], breg[4 * c + t], acc);
    }
}
__device__ __forceinline__ void zero16(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
}

__global__ __launch_bounds__(448) void attn_fwd_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ out,
                                                           float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = reinterpret_cast<float*>(smem + F_IMG);   // row stride 64: only read down columns
    const int bh = blockIdx.x, b = bh / NH, h = bh - b * NH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_rows_f32<448, FLD>(k + (size_t)bh * NT * HD, HD, Ks, tid);
    stage_rows_f32<448, HD>(v + (size_t)bh * NT * HD, HD, Vs, tid);
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31, qr = min(qrow, NT - 1);
    float qf[32];
    load_half_row(q + ((size_t)bh * NT + qr) * HD + hi * 32, qf);
    f32x16 st[7];
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
        zero16(st[kt]);
        mma_rowfrag(st[kt], Ks, kt * 32, l31, hi, qf);   // S^T[key][q]
    }
    // st[kt][r] = S^T[key = kt*32 + (r&3) + 8*(r>>2) + 4*hi][q = l31]; mask the pad keys
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = 192 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= NT) st[6][r] = -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, st[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(st[kt][r] - m);
            st[kt][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    if (hi == 0 && qrow < NT) lse[(size_t)bh * NT + qrow] = m + logf(sum);

    f32x16 o[2];
    zero16(o[0]); zero16(o[1]);
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // O^T[d][q] += V^T[d][key] P^T[key][q]
            const float* vp = Vs + (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * HD + l31;
            o[0] = MFMA32F(vp[0], st[kt][r], o[0]);
            o[1] = MFMA32F(vp[32], st[kt][r], o[1]);
        }
    if (qrow < NT) {
        const float inv = 1.0f / sum;
        float* op = out + ((size_t)b * NT + qrow) * D + h * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store4(op + dt * 32 + 8 * g + 4 * hi, o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                       o[dt][4 * g + 3] * inv);
    }
}

__global__ __launch_bounds__(448) void attn_bwd_dq_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ o,
                                                              const float* __restrict__ dout,
                                                              const float* __restrict__ lse, float* __restrict__ delta,
                                                              float* __restrict__ dqkv, bf16* __restrict__ dqkv3, float s3) {
    // dqkv3 (fp32 split form): the result times s3 goes out as the 16-bit hi / hi / lo operand of the qkv dgrad GEMM ([M, 3 * 2304])
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = reinterpret_cast<float*>(smem + F_IMG);
    const int bh = blockIdx.x, b = bh / NH, h = bh - b * NH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_rows_f32<448, FLD>(k + (size_t)bh * NT * HD, HD, Ks, tid);
    stage_rows_f32<448, FLD>(v + (size_t)bh * NT * HD, HD, Vs, tid);
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31, qr = min(qrow, NT - 1);
    const size_t trow = ((size_t)b * NT + qr) * D + h * HD + hi * 32;
    float qf[32], dof[32];
    load_half_row(q + ((size_t)bh * NT + qr) * HD + hi * 32, qf);
    load_half_row(dout + trow, dof);
    float dl = 0.f;
    {
        float of[32];
        load_half_row(o + trow, of);
#pragma unroll
        for (int d = 0; d < 32; ++d) dl = fmaf(dof[d], of[d], dl);
    }
    dl += __shfl_xor(dl, 32, 64);
    const float L = lse[(size_t)bh * NT + qr];
    if (hi == 0 && qrow < NT) delta[(size_t)bh * NT + qrow] = dl;

    f32x16 dq[2];
    zero16(dq[0]); zero16(dq[1]);
#pragma unroll 1
    for (int kt = 0; kt < 7; ++kt) {
        f32x16 s, dp;
        zero16(s); zero16(dp);
        mma_rowfrag(s, Ks, kt * 32, l31, hi, qf);     // S^T[key][q]
        mma_rowfrag(dp, Vs, kt * 32, l31, hi, dof);   // dP^T[key][q]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            s[r] = key < NT ? expf(s[r] - L) * (dp[r] - dl) : 0.f;   // dS^T
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            const float* kp = Ks + (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * FLD + l31;
            dq[0] = MFMA32F(kp[0], s[r], dq[0]);
            dq[1] = MFMA32F(kp[32], s[r], dq[1]);
        }
    }
    if (qrow < NT && dqkv3) {
        bf16* op3 = dqkv3 + ((size_t)b * NT + qrow) * (SPLIT_A * 3 * D) + h * HD;
        const float sc = 0.125f * s3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store4_split3(op3 + dt * 32 + 8 * g + 4 * hi, 3 * D, dq[dt][4 * g] * sc, dq[dt][4 * g + 1] * sc, dq[dt][4 * g + 2] * sc,
                              dq[dt][4 * g + 3] * sc);
    } else if (qrow < NT) {
        float* op = dqkv + ((size_t)b * NT + qrow) * (3 * D) + h * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store4(op + dt * 32 + 8 * g + 4 * hi, dq[dt][4 * g] * 0.125f, dq[dt][4 * g + 1] * 0.125f,
                       dq[dt][4 * g + 2] * 0.125f, dq[dt][4 * g + 3] * 0.125f);
    }
}

// dK, dV per key tile (7 waves, one 32-key tile each)
__global__ __launch_bounds__(448) void attn_bwd_dkv_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v,
                                                               const float* __restrict__ dout,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               float* __restrict__ dqkv, bf16* __restrict__ dqkv3, float s3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Qs = reinterpret_cast<float*>(smem);
    float* dOs = reinterpret_cast<float*>(smem + F_IMG);
    float* lse_s = reinterpret_cast<float*>(smem + 2 * F_IMG);
    float* del_s = lse_s + NPAD;
    const int bh = blockIdx.x, b = bh / NH, h = bh - b * NH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_rows_f32<448, FLD>(q + (size_t)bh * NT * HD, HD, Qs, tid);
    stage_rows_f32<448, FLD>(dout + (size_t)b * NT * D + h * HD, D, dOs, tid);
    if (tid < NPAD) {
        lse_s[tid] = tid < NT ? lse[(size_t)bh * NT + tid] : 0.f;
        del_s[tid] = tid < NT ? delta[(size_t)bh * NT + tid] : 0.f;
    }
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5;
    const int key = wave * 32 + l31, kr = min(key, NT - 1);
    float kf[32], vf[32];
    load_half_row(k + ((size_t)bh * NT + kr) * HD + hi * 32, kf);
    load_half_row(v + ((size_t)bh * NT + kr) * HD + hi * 32, vf);
    f32x16 aK[2], aV[2];
    zero16(aK[0]); zero16(aK[1]); zero16(aV[0]); zero16(aV[1]);
#pragma unroll 1
    for (int qt = 0; qt < 7; ++qt) {
        f32x16 s, dp;
        zero16(s); zero16(dp);
        mma_rowfrag(s, Qs, qt * 32, l31, hi, kf);     // S[q][key]  (rows q in registers, column key = lane)
        mma_rowfrag(dp, dOs, qt * 32, l31, hi, vf);   // dP[q][key]
        f32x16 p;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q0 = qt * 32 + 8 * g + 4 * hi;
            const float4 L4 = *reinterpret_cast<const float4*>(lse_s + q0);
            const float4 D4 = *reinterpret_cast<const float4*>(del_s + q0);
            const float Ls[4] = {L4.x, L4.y, L4.z, L4.w};
            const float Ds[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const bool ok = (q0 + e < NT) && (key < NT);
                const float pv = ok ? expf(s[r] - Ls[e]) : 0.f;
                p[r] = pv;
                s[r] = ok ? pv * (dp[r] - Ds[e]) : 0.f;  // dS
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float* dop = dOs + qi * FLD + l31;
            const float* qp = Qs + qi * FLD + l31;
            aV[0] = MFMA32F(dop[0], p[r], aV[0]);    // dV^T[d][key] += dO^T[d][q] P[q][key]
            aV[1] = MFMA32F(dop[32], p[r], aV[1]);
            aK[0] = MFMA32F(qp[0], s[r], aK[0]);     // dK^T[d][key] += Q^T[d][q] dS[q][key]
            aK[1] = MFMA32F(qp[32], s[r], aK[1]);
        }
    }
    if (key < NT && dqkv3) {
        bf16* op3 = dqkv3 + ((size_t)b * NT + key) * (SPLIT_A * 3 * D) + h * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hi;
                store4_split3(op3 + D + d, 3 * D, aK[dt][4 * g] * s3, aK[dt][4 * g + 1] * s3, aK[dt][4 * g + 2] * s3, aK[dt][4 * g + 3] * s3);
                store4_split3(op3 + 2 * D + d, 3 * D, aV[dt][4 * g] * s3, aV[dt][4 * g + 1] * s3, aV[dt][4 * g + 2] * s3, aV[dt][4 * g + 3] * s3);
            }
    } else if (key < NT) {
        float* op = dqkv + ((size_t)b * NT + key) * (3 * D) + h * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hi;
                store4(op + D + d, aK[dt][4 * g], aK[dt][4 * g + 1], aK[dt][4 * g + 2], aK[dt][4 * g + 3]);
                store4(op + 2 * D + d, aV[dt][4 * g], aV[dt][4 * g + 1], aV[dt][4 * g + 2], aV[dt][4 * g + 3]);
            }
    }
}

// ------------------------------------------------------------------------------------------
// forward, fp32 operands as 16-bit hi / lo parts (the fp32 mode's DYT_OPT_F32_SPLIT16 form)
// ------------------------------------------------------------------------------------------
// The tiling, layouts and lane mapping of attn_fwd_bf16_kernel with every product computed as hi*hi + hi*lo + lo*hi of two
// 16-bit parts per fp32 operand (x = hi + lo, hi = rn16(x), lo = rn16(x - hi); the dropped lo*lo term is 2^-22 relative): K as
// two row images, V^T as two transposed images (122.9 KB of LDS), q split in registers, the probabilities split per key tile
// right where they are packed.  fp32 accumulation, fp32 softmax statistics, fp32 output.  Persistent over heads, no
// cross-head prefetch (the fp32 rows would double the staging registers).
// exp(x) for the split kernels: v_exp_f32 of the compensated product x * log2(e) -- h = rn(x * L2E), l = the product's remainder plus
// x times the low part of log2(e), exp(x) = 2^h (1 + l ln 2) -- six instructions instead of libm expf's ~25, within ~1.5 ulp
// (a bare __expf is off by |x| 2^-24 relative: 2e-6 at the scores' range, the level of the whole split mode's error)
__device__ __forceinline__ float exp_c(float x) {
    x = fmaxf(x, -104.0f);   // masked scores (-inf, -1e30) would make h - h a NaN; exp(-104) is 0 in fp32 anyway
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.925962991e-8f;
    const float h = x * L2E;
    const float l = __fmaf_rn(x, L2E_LO, __fmaf_rn(x, L2E, -h));
    const float e = __builtin_amdgcn_exp2f(h);
    return __fmaf_rn(e * 0.693147182464599609375f, l, e);
}
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const Split2 s0 = split2(x0[i]), s1 = split2(x1[i]);
        hi[i] = s0.hi; lo[i] = s0.lo;
        hi[4 + i] = s1.hi; lo[4 + i] = s1.lo;
    }
}
// PARTS = 1: the hi * hi product alone for S and P V (IEEE-half attention on the same planes, fp32 softmax): the complete_model pass of
// "fp16x3q", whose output no token-keep decision depends on (its budget is the 1e-3 logit bar, not the gate's 1e-5)
template <bool PLANES, int PARTS = 3>   // PLANES: q / k / v arrive as 16-bit hi (q16 / k16 / v16) + lo planes written by the QKV epilogue: staged without conversion
__global__ __launch_bounds__(448) void attn_fwd_split_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, float* __restrict__ out,
                                                             float* __restrict__ lse, int nheads, bf16* __restrict__ out3,
                                                             bf16* __restrict__ q16, bf16* __restrict__ k16, bf16* __restrict__ v16,
                                                             bf16* __restrict__ o16, int f8, const bf16* __restrict__ qlo,
                                                             const bf16* __restrict__ klo, const bf16* __restrict__ vlo) {
    // q16 / k16 / v16 / o16 (optional): the hi parts = the 16-bit roundings of q, k, v and the output, saved for a backward pass that
    // runs on 16-bit operands ("fp16x3h"); `out` may then be null (the proj GEMM reads out3)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Kh = reinterpret_cast<bf16*>(smem);
    bf16* Kl = reinterpret_cast<bf16*>(smem + ROW_IMG);
    bf16* Vth = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG);
    bf16* Vtl = reinterpret_cast<bf16*>(smem + 2 * ROW_IMG + TR_IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31, qr = min(qrow, NT - 1);
    for (int bh = blockIdx.x; bh < nheads; bh += gridDim.x) {
        const int b = bh / NH, h = bh - b * NH;
        // ---- stage K (two row images) and V (two transposed images): task = (row pair, 8-column chunk), 896 tasks / 448 threads
        const float* kb = k + (size_t)bh * NT * HD;
        const float* vb = v + (size_t)bh * NT * HD;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + u * 448, pr = t >> 3, c = t & 7, r0 = pr * 2;
            bf16x8 ah, al, bhh, bl, vah, val, vbh, vbl;
            if constexpr (PLANES) {
                const size_t o0 = ((size_t)bh * NT + min(r0, NT - 1)) * HD + c * 8, o1 = ((size_t)bh * NT + min(r0 + 1, NT - 1)) * HD + c * 8;
                ah = *reinterpret_cast<const bf16x8*>(k16 + o0); bhh = *reinterpret_cast<const bf16x8*>(k16 + o1);
                vah = *reinterpret_cast<const bf16x8*>(v16 + o0); vbh = *reinterpret_cast<const bf16x8*>(v16 + o1);
                if constexpr (PARTS == 3) {
                    al = *reinterpret_cast<const bf16x8*>(klo + o0); bl = *reinterpret_cast<const bf16x8*>(klo + o1);
                    val = *reinterpret_cast<const bf16x8*>(vlo + o0); vbl = *reinterpret_cast<const bf16x8*>(vlo + o1);
                } else {
                    al = zero8(); bl = zero8(); val = zero8(); vbl = zero8();
                }
            } else {
            const float* k0 = kb + (size_t)min(r0, NT - 1) * HD + c * 8;
            const float* k1 = kb + (size_t)min(r0 + 1, NT - 1) * HD + c * 8;
            const float* v0 = vb + (size_t)min(r0, NT - 1) * HD + c * 8;
            const float* v1 = vb + (size_t)min(r0 + 1, NT - 1) * HD + c * 8;
            const f32x4 ka0 = *reinterpret_cast<const f32x4*>(k0), ka1 = *reinterpret_cast<const f32x4*>(k0 + 4);
            const f32x4 kb0 = *reinterpret_cast<const f32x4*>(k1), kb1 = *reinterpret_cast<const f32x4*>(k1 + 4);
            const f32x4 va0 = *reinterpret_cast<const f32x4*>(v0), va1 = *reinterpret_cast<const f32x4*>(v0 + 4);
            const f32x4 vb0 = *reinterpret_cast<const f32x4*>(v1), vb1 = *reinterpret_cast<const f32x4*>(v1 + 4);
            split8(ka0, ka1, ah, al); split8(kb0, kb1, bhh, bl);
            split8(va0, va1, vah, val); split8(vb0, vb1, vbh, vbl);
            }
            if (r0 >= NT) { ah = zero8(); al = zero8(); }
            if (r0 + 1 >= NT) { bhh = zero8(); bl = zero8(); }
            *reinterpret_cast<bf16x8*>(Kh + r0 * RLD + c * 8) = ah;
            *reinterpret_cast<bf16x8*>(Kh + (r0 + 1) * RLD + c * 8) = bhh;
            if constexpr (PARTS == 3) {
                *reinterpret_cast<bf16x8*>(Kl + r0 * RLD + c * 8) = al;
                *reinterpret_cast<bf16x8*>(Kl + (r0 + 1) * RLD + c * 8) = bl;
            }
            if (!PLANES && k16) {
                if (r0 < NT) *reinterpret_cast<bf16x8*>(k16 + ((size_t)bh * NT + r0) * HD + c * 8) = ah;
                if (r0 + 1 < NT) *reinterpret_cast<bf16x8*>(k16 + ((size_t)bh * NT + r0 + 1) * HD + c * 8) = bhh;
            }
            ah = vah; al = val; bhh = vbh; bl = vbl;
            if (r0 >= NT) { ah = zero8(); al = zero8(); }
            if (r0 + 1 >= NT) { bhh = zero8(); bl = zero8(); }
            if (!PLANES && v16) {
                if (r0 < NT) *reinterpret_cast<bf16x8*>(v16 + ((size_t)bh * NT + r0) * HD + c * 8) = ah;
                if (r0 + 1 < NT) *reinterpret_cast<bf16x8*>(v16 + ((size_t)bh * NT + r0 + 1) * HD + c * 8) = bhh;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x2 ph = {ah[i], bhh[i]}, pl = {al[i], bl[i]};
                *reinterpret_cast<bf16x2*>(Vth + (c * 8 + i) * TLD + r0) = ph;
                if constexpr (PARTS == 3) *reinterpret_cast<bf16x2*>(Vtl + (c * 8 + i) * TLD + r0) = pl;
            }
        }
        // ---- this wave's 32 query rows, split
        bf16x8 qh[4], ql[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (PLANES) {
                const size_t o = ((size_t)bh * NT + qr) * HD + ks * 16 + hi * 8;
                qh[ks] = *reinterpret_cast<const bf16x8*>(q16 + o);
                if constexpr (PARTS == 3) ql[ks] = *reinterpret_cast<const bf16x8*>(qlo + o);
            } else {
            const float* qp = q + ((size_t)bh * NT + qr) * HD + ks * 16 + hi * 8;
            split8(*reinterpret_cast<const f32x4*>(qp), *reinterpret_cast<const f32x4*>(qp + 4), qh[ks], ql[ks]);
            if (q16 && qrow < NT) *reinterpret_cast<bf16x8*>(q16 + ((size_t)bh * NT + qrow) * HD + ks * 16 + hi * 8) = qh[ks];
            }
        }
        __syncthreads();
        f32x16 st[7];
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 7; ++kt) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Kh + (kt * 32 + l31) * RLD + ks * 16 + hi * 8);
                if constexpr (PARTS == 3) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Kl + (kt * 32 + l31) * RLD + ks * 16 + hi * 8);
                    st[kt] = MFMA32(al, qh[ks], st[kt]);   // small terms first
                    st[kt] = MFMA32(ah, ql[ks], st[kt]);
                }
                st[kt] = MFMA32(ah, qh[ks], st[kt]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 192 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= NT) st[6][r] = -INFINITY;
        }
        float mp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < 7; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mp[r & 3] = fmaxf(mp[r & 3], st[kt][r]);
        float m = fmaxf(fmaxf(mp[0], mp[1]), fmaxf(mp[2], mp[3]));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sp[4] = {0.f, 0.f, 0.f, 0.f};
        f32x16 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < 7; ++kt) {
            bf16x8 vh[2][2], vl[2][2];
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    vh[half][dt] = join44(Vth + (dt * 32 + l31) * TLD + kt * 32 + half * 16 + 4 * hi);
                    if constexpr (PARTS == 3) vl[half][dt] = join44(Vtl + (dt * 32 + l31) * TLD + kt * 32 + half * 16 + 4 * hi);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (one-part form: a bare v_exp_f32 -- its |x| 2^-24 relative error is far below the half rounding of p)
                const float p = PARTS == 3 ? exp_c(st[kt][r] - m) : __builtin_amdgcn_exp2f((st[kt][r] - m) * 1.44269502162933349609375f);
                st[kt][r] = p;
                sp[r & 3] += p;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf16x8 ph, pl;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const Split2 sp2 = split2(st[kt][half * 8 + i]);
                    ph[i] = sp2.hi; pl[i] = sp2.lo;
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    if constexpr (PARTS == 3) {
                        o[dt] = MFMA32(vl[half][dt], ph, o[dt]);
                        o[dt] = MFMA32(vh[half][dt], pl, o[dt]);
                    }
                    o[dt] = MFMA32(vh[half][dt], ph, o[dt]);
                }
            }
        }
        float sum = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        sum += __shfl_xor(sum, 32, 64);
        if (hi == 0 && qrow < NT) lse[(size_t)bh * NT + qrow] = m + logf(sum);
        if (qrow < NT) {
            const float inv = 1.0f / sum;
            float* op = out ? out + ((size_t)b * NT + qrow) * D + h * HD : nullptr;
            bf16* op16 = o16 ? o16 + ((size_t)b * NT + qrow) * D + h * HD : nullptr;
            bf16* op3 = out3 ? out3 + ((size_t)b * NT + qrow) * (SPLIT_A * D) + h * HD : nullptr;   // + the split operand of the proj GEMM
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * hi;
                    const float o0 = o[dt][4 * g] * inv, o1 = o[dt][4 * g + 1] * inv, o2 = o[dt][4 * g + 2] * inv, o3 = o[dt][4 * g + 3] * inv;
                    if (op) store4(op + d, o0, o1, o2, o3);
                    if (op16) store4(op16 + d, o0, o1, o2, o3);
                    if (op3) { if (f8) store4_split_f8(op3 - h * HD, D, h * HD + d, o0, o1, o2, o3); else store4_split3(op3 + d, D, o0, o1, o2, o3); }
                }
        }
        __syncthreads();   // every wave is done with this head's images before they are overwritten
    }
}

// ------------------------------------------------------------------------------------------
// backward, fp32 operands as 16-bit hi / lo parts (DYT_OPT_F32_SPLIT16)
// ------------------------------------------------------------------------------------------
// The two-kernel backward of the bf16 path (dQ per query tile, dK/dV per key tile, lane mapping and fragment layouts unchanged)
// with every product as hi*hi + hi*lo + lo*hi.  Every LDS image exists twice (hi, lo), so a head is staged in two HALVES of
// 128 rows (keys for dQ: 108 KB; queries for dK/dV: 142 KB); the accumulators live across the halves.  fp32 in, fp32
// statistics (lse, delta), output either fp32 or -- times s3 -- directly as the split operand of the qkv dgrad GEMM.
// dO is multiplied by gs (a power of two) before it is split, which carries through dP, dS, dQ, dK, dV: gradient-sized values
// would otherwise put their lo parts (and many hi parts) into the fp16 subnormals; the outputs are divided by gs again.
constexpr int HROWS = 128;                 // rows staged per half (4 tiles of 32; the second half holds tiles 4..6)
constexpr int TLDH = HROWS + 4;            // 16-bit elements per row of a transposed half image (264 B: conflict-free b64 reads)
constexpr int ROW_H = HROWS * RLD * 2;     // 18432 B
constexpr int TR_H = HD * TLDH * 2;        // 16896 B

// rows [row0, row0 + 128) of a [197][64] fp32 matrix (row stride ld) -> hi / lo row images and / or hi / lo transposed images
__device__ __forceinline__ void stage_split_half(const float* __restrict__ src, size_t ld, int row0, bf16* rH, bf16* rL, bf16* tH,
                                                 bf16* tL, int tid, float scale = 1.0f) {
    for (int t = tid; t < (HROWS / 2) * 8; t += 448) {
        const int pr = t >> 3, c = t & 7, r0 = pr * 2, g0 = row0 + r0;
        const float* p0 = src + (size_t)min(g0, NT - 1) * ld + c * 8;
        const float* p1 = src + (size_t)min(g0 + 1, NT - 1) * ld + c * 8;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(p0) * scale, a1 = *reinterpret_cast<const f32x4*>(p0 + 4) * scale;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p1) * scale, b1 = *reinterpret_cast<const f32x4*>(p1 + 4) * scale;
        bf16x8 ah, al, bh, bl;
        split8(a0, a1, ah, al); split8(b0, b1, bh, bl);
        if (g0 >= NT) { ah = zero8(); al = zero8(); }
        if (g0 + 1 >= NT) { bh = zero8(); bl = zero8(); }
        if (rH) {
            *reinterpret_cast<bf16x8*>(rH + r0 * RLD + c * 8) = ah;
            *reinterpret_cast<bf16x8*>(rH + (r0 + 1) * RLD + c * 8) = bh;
        }
        if (rL) {   // (null where only one-part products read the image: the lo half is never looked at)
            *reinterpret_cast<bf16x8*>(rL + r0 * RLD + c * 8) = al;
            *reinterpret_cast<bf16x8*>(rL + (r0 + 1) * RLD + c * 8) = bl;
        }
        if (tH) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x2 ph = {ah[i], bh[i]};
                *reinterpret_cast<bf16x2*>(tH + (c * 8 + i) * TLDH + r0) = ph;
            }
        }
        if (tL) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x2 pl = {al[i], bl[i]};
                *reinterpret_cast<bf16x2*>(tL + (c * 8 + i) * TLDH + r0) = pl;
            }
        }
    }
}
__device__ __forceinline__ void split_pack8(const f32x16& v, int base, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { const Split2 s2 = split2(v[base + i]); hi[i] = s2.hi; lo[i] = s2.lo; }
}
// acc += A B with A = (ah, al), B = (bh, bl): small terms first
#define MFMA3(acc, ah, al, bh, bl) do { acc = MFMA32(al, bh, acc); acc = MFMA32(ah, bl, acc); acc = MFMA32(ah, bh, acc); } while (0)
// the gradient products (dP, dQ, dK, dV) of the backward kernels: GP = 3 as above, GP = 1 the hi * hi product alone (the score
// recomputation always takes three: an error of S is an error of the exponent)
#define MFMAG(acc, ah, al, bh, bl) do { if constexpr (GP == 3) MFMA3(acc, ah, al, bh, bl); else acc = MFMA32(ah, bh, acc); } while (0)

template <int GP>
__global__ __launch_bounds__(448) void attn_bwd_dq_split_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ o,
                                                                const float* __restrict__ dout, const float* __restrict__ lse,
                                                                float* __restrict__ delta, float* __restrict__ dqkv,
                                                                bf16* __restrict__ dqkv3, float s3, int nheads, float gs, int hi_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Kh = reinterpret_cast<bf16*>(smem);
    bf16* Kl = reinterpret_cast<bf16*>(smem + ROW_H);
    bf16* Vh = reinterpret_cast<bf16*>(smem + 2 * ROW_H);
    bf16* Vl = reinterpret_cast<bf16*>(smem + 3 * ROW_H);
    bf16* Kth = reinterpret_cast<bf16*>(smem + 4 * ROW_H);
    bf16* Ktl = reinterpret_cast<bf16*>(smem + 4 * ROW_H + TR_H);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31, qr = min(qrow, NT - 1);
    for (int bh = blockIdx.x; bh < nheads; bh += gridDim.x) {
        const int b = bh / NH, h = bh - b * NH;
        bf16x8 qh[4], ql[4], doh[4], dol[4];
        float dl = 0.f;
        {
            const float* qp = q + ((size_t)bh * NT + qr) * HD + hi * 8;
            const size_t trow = ((size_t)b * NT + qr) * D + h * HD + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(qp + ks * 16), q1 = *reinterpret_cast<const f32x4*>(qp + ks * 16 + 4);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(dout + trow + ks * 16), d1 = *reinterpret_cast<const f32x4*>(dout + trow + ks * 16 + 4);
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(o + trow + ks * 16), o1 = *reinterpret_cast<const f32x4*>(o + trow + ks * 16 + 4);
                split8(q0, q1, qh[ks], ql[ks]);
                split8(d0 * gs, d1 * gs, doh[ks], dol[ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) dl += d0[i] * o0[i] + d1[i] * o1[i];
            }
        }
        const float L = lse[(size_t)bh * NT + qr];
        dl += __shfl_xor(dl, 32, 64);
        if (hi == 0 && qrow < NT) delta[(size_t)bh * NT + qrow] = dl;
        const float dlg = dl * gs;
        f32x16 dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            __syncthreads();   // the previous half's (head's) images are no longer read
            stage_split_half(k + (size_t)bh * NT * HD, HD, hf * HROWS, Kh, Kl, Kth, GP == 3 ? Ktl : nullptr, tid);
            stage_split_half(v + (size_t)bh * NT * HD, HD, hf * HROWS, Vh, GP == 3 ? Vl : nullptr, nullptr, nullptr, tid);
            __syncthreads();
            const int ntile = hf == 0 ? 4 : 3;
#pragma unroll 1
            for (int t = 0; t < ntile; ++t) {
                f32x16 s_, dp_;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp_[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int off = (t * 32 + l31) * RLD + ks * 16 + hi * 8;
                    const bf16x8 kah = *reinterpret_cast<const bf16x8*>(Kh + off), kal = *reinterpret_cast<const bf16x8*>(Kl + off);
                    const bf16x8 vah = *reinterpret_cast<const bf16x8*>(Vh + off), val = *reinterpret_cast<const bf16x8*>(Vl + off);
                    MFMA3(s_, kah, kal, qh[ks], ql[ks]);
                    MFMAG(dp_, vah, val, doh[ks], dol[ks]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (hf * 4 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float p = key < NT ? exp_c(s_[r] - L) : 0.f;
                    s_[r] = p * (dp_[r] - dlg);       // gs * dS^T
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    bf16x8 dsh, dsl;
                    split_pack8(s_, half * 8, dsh, dsl);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int toff = (dt * 32 + l31) * TLDH + t * 32 + half * 16 + 4 * hi;
                        const bf16x8 kth = join44(Kth + toff), ktl = join44(Ktl + toff);
                        MFMAG(dq[dt], kth, ktl, dsh, dsl);   // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
                    }
                }
            }
        }
        if (qrow < NT) {
            const float sc = 0.125f * (dqkv3 ? s3 : 1.0f) / gs;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * hi;
                    if (dqkv3) store4_split3(dqkv3 + ((size_t)b * NT + qrow) * (SPLIT_A * 3 * D) + h * HD + d, 3 * D, dq[dt][4 * g] * sc, dq[dt][4 * g + 1] * sc,
                                             dq[dt][4 * g + 2] * sc, dq[dt][4 * g + 3] * sc, hi_only != 0);
                    else store4(dqkv + ((size_t)b * NT + qrow) * (3 * D) + h * HD + d, dq[dt][4 * g] * sc, dq[dt][4 * g + 1] * sc,
                                dq[dt][4 * g + 2] * sc, dq[dt][4 * g + 3] * sc);
                }
        }
    }
}

template <int GP>
__global__ __launch_bounds__(448) void attn_bwd_dkv_split_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const float* __restrict__ dout,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                                 float* __restrict__ dqkv, bf16* __restrict__ dqkv3, float s3, int nheads, float gs, int hi_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Qh = reinterpret_cast<bf16*>(smem);
    bf16* Ql = reinterpret_cast<bf16*>(smem + ROW_H);
    bf16* Dh = reinterpret_cast<bf16*>(smem + 2 * ROW_H);
    bf16* Dl = reinterpret_cast<bf16*>(smem + 3 * ROW_H);
    bf16* Qth = reinterpret_cast<bf16*>(smem + 4 * ROW_H);
    bf16* Qtl = reinterpret_cast<bf16*>(smem + 4 * ROW_H + TR_H);
    bf16* Dth = reinterpret_cast<bf16*>(smem + 4 * ROW_H + 2 * TR_H);
    bf16* Dtl = reinterpret_cast<bf16*>(smem + 4 * ROW_H + 3 * TR_H);
    float* lse_s = reinterpret_cast<float*>(smem + 4 * ROW_H + 4 * TR_H);
    float* del_s = lse_s + HROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int key = wave * 32 + l31, kr_ = min(key, NT - 1);
    for (int bh = blockIdx.x; bh < nheads; bh += gridDim.x) {
        const int b = bh / NH, h = bh - b * NH;
        bf16x8 kh[4], kl[4], vh[4], vl[4];
        {
            const float* kp = k + ((size_t)bh * NT + kr_) * HD + hi * 8;
            const float* vp = v + ((size_t)bh * NT + kr_) * HD + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                split8(*reinterpret_cast<const f32x4*>(kp + ks * 16), *reinterpret_cast<const f32x4*>(kp + ks * 16 + 4), kh[ks], kl[ks]);
                split8(*reinterpret_cast<const f32x4*>(vp + ks * 16), *reinterpret_cast<const f32x4*>(vp + ks * 16 + 4), vh[ks], vl[ks]);
            }
        }
        f32x16 aK[2], aV[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { aK[0][r] = 0.f; aK[1][r] = 0.f; aV[0][r] = 0.f; aV[1][r] = 0.f; }
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            __syncthreads();
            stage_split_half(q + (size_t)bh * NT * HD, HD, hf * HROWS, Qh, Ql, Qth, GP == 3 ? Qtl : nullptr, tid);
            stage_split_half(dout + (size_t)b * NT * D + h * HD, D, hf * HROWS, Dh, GP == 3 ? Dl : nullptr, Dth, GP == 3 ? Dtl : nullptr, tid, gs);
            if (tid < HROWS) {
                const int qg = hf * HROWS + tid;
                lse_s[tid] = qg < NT ? lse[(size_t)bh * NT + qg] : 0.f;
                del_s[tid] = qg < NT ? delta[(size_t)bh * NT + qg] * gs : 0.f;
            }
            __syncthreads();
            const int ntile = hf == 0 ? 4 : 3;
#pragma unroll 1
            for (int t = 0; t < ntile; ++t) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int off = (t * 32 + l31) * RLD + ks * 16 + hi * 8;
                    const bf16x8 qah = *reinterpret_cast<const bf16x8*>(Qh + off), qal = *reinterpret_cast<const bf16x8*>(Ql + off);
                    const bf16x8 dah = *reinterpret_cast<const bf16x8*>(Dh + off), dal = *reinterpret_cast<const bf16x8*>(Dl + off);
                    MFMA3(s, qah, qal, kh[ks], kl[ks]);      // S[q][key]
                    MFMAG(dp, dah, dal, vh[ks], vl[ks]);     // dP[q][key]
                }
                f32x16 p;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q0 = t * 32 + 8 * g + 4 * hi;   // row inside the half
                    const float4 L4 = *reinterpret_cast<const float4*>(lse_s + q0);
                    const float4 D4 = *reinterpret_cast<const float4*>(del_s + q0);
                    const float Ls[4] = {L4.x, L4.y, L4.z, L4.w};
                    const float Ds[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const bool ok = (hf * HROWS + q0 + e < NT) && (key < NT);
                        const float pv = ok ? exp_c(s[r] - Ls[e]) : 0.f;
                        p[r] = pv;
                        s[r] = pv * (dp[r] - Ds[e]);  // dS
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    bf16x8 ph, pl, dsh, dsl;
                    split_pack8(p, half * 8, ph, pl);
                    split_pack8(s, half * 8, dsh, dsl);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int toff = (dt * 32 + l31) * TLDH + t * 32 + half * 16 + 4 * hi;
                        const bf16x8 doth = join44(Dth + toff), dotl = join44(Dtl + toff);
                        const bf16x8 qth = join44(Qth + toff), qtl = join44(Qtl + toff);
                        MFMAG(aV[dt], doth, dotl, ph, pl);    // dV^T[d][key] += dO^T[d][q] P[q][key]
                        MFMAG(aK[dt], qth, qtl, dsh, dsl);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
                    }
                }
            }
        }
        if (key < NT) {
            const float sc = (dqkv3 ? s3 : 1.0f) / gs;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * hi;
                    if (dqkv3) {
                        bf16* op3 = dqkv3 + ((size_t)b * NT + key) * (SPLIT_A * 3 * D) + h * HD + d;
                        store4_split3(op3 + D, 3 * D, aK[dt][4 * g] * sc, aK[dt][4 * g + 1] * sc, aK[dt][4 * g + 2] * sc, aK[dt][4 * g + 3] * sc, hi_only != 0);
                        store4_split3(op3 + 2 * D, 3 * D, aV[dt][4 * g] * sc, aV[dt][4 * g + 1] * sc, aV[dt][4 * g + 2] * sc, aV[dt][4 * g + 3] * sc, hi_only != 0);
                    } else {
                        float* op = dqkv + ((size_t)b * NT + key) * (3 * D) + h * HD + d;
                        store4(op + D, aK[dt][4 * g] * sc, aK[dt][4 * g + 1] * sc, aK[dt][4 * g + 2] * sc, aK[dt][4 * g + 3] * sc);
                        store4(op + 2 * D, aV[dt][4 * g] * sc, aV[dt][4 * g + 1] * sc, aV[dt][4 * g + 2] * sc, aV[dt][4 * g + 3] * sc);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
static int set_lds(const void* f, size_t bytes) {
    DYT_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
// the attribute belongs to the (kernel, device) pair: one flag per device and kernel family
static bool* attr_flag(int family) {
    static bool done[3][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return &done[family][dev & 63];
}

static int g_attn_f32_split = 0;   // process-wide: the split forward kernel for every fp32-mode call (unit entries; contexts pass their own flag)
void set_attn_f32_split(int on) { g_attn_f32_split = on; }

int launch_attn_fwd(int precision, const void* q, const void* k, const void* v, void* out, float* lse, int batch,
                    hipStream_t s, int split16, void* out3, const AttnSave16* save16, int out3_f8, int parts) {
    const int grid = batch * NH;
    if (dbg_skip(2)) return 0;
    if ((save16 || !out) && !(precision == 0 && (split16 || g_attn_f32_split))) { set_error("attention forward: 16-bit copies / no fp32 output need the split kernel"); return -1; }
    if (!out && !out3) { set_error("attention forward: no output"); return -1; }
    if (precision == 0 && (split16 || g_attn_f32_split)) {
        const size_t lds = 2 * ROW_IMG + 2 * TR_IMG;
        static bool done[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (!done[dev & 63]) {
            if (set_lds((const void*)attn_fwd_split_kernel<false>, lds) || set_lds((const void*)attn_fwd_split_kernel<true>, lds) ||
                set_lds((const void*)attn_fwd_split_kernel<true, 1>, lds)) return -2;
            done[dev & 63] = true;
        }
        if (parts != 3 && !(save16 && save16->q_lo && parts == 1)) { set_error("attention forward: the one-part form needs the planar q / k / v"); return -1; }
        // one-part form with nothing but the operand image to write (every block of a complete_model pass but the cls-only last one): the
        // round-5 16-bit kernel on the hi planes, its result split on the way out (DYT_OPT_ATTN_V2 bit 0)
        if (save16 && save16->q_lo && parts == 1 && !out && out3 && out3_f8 && !save16->o && (get_attn_v2() & 1))
            return launch_attn_fwd_v2(save16->q, save16->k, save16->v, out3, lse, batch, s, 1);
        if (save16 && save16->q_lo && parts == 1)
            hipLaunchKernelGGL((attn_fwd_split_kernel<true, 1>), dim3(min(grid, 256)), dim3(448), lds, s, nullptr, nullptr, nullptr, (float*)out, lse, grid,
                               (bf16*)out3, (bf16*)save16->q, (bf16*)save16->k, (bf16*)save16->v, (bf16*)save16->o, out3_f8,
                               (const bf16*)save16->q_lo, (const bf16*)save16->k_lo, (const bf16*)save16->v_lo);
        else if (save16 && save16->q_lo)
            hipLaunchKernelGGL(attn_fwd_split_kernel<true>, dim3(min(grid, 256)), dim3(448), lds, s, nullptr, nullptr, nullptr, (float*)out, lse, grid,
                               (bf16*)out3, (bf16*)save16->q, (bf16*)save16->k, (bf16*)save16->v, (bf16*)save16->o, out3_f8,
                               (const bf16*)save16->q_lo, (const bf16*)save16->k_lo, (const bf16*)save16->v_lo);
        else
        hipLaunchKernelGGL(attn_fwd_split_kernel<false>, dim3(min(grid, 256)), dim3(448), lds, s, (const float*)q, (const float*)k,
                           (const float*)v, (float*)out, lse, grid, (bf16*)out3, save16 ? (bf16*)save16->q : nullptr, save16 ? (bf16*)save16->k : nullptr,
                           save16 ? (bf16*)save16->v : nullptr, save16 ? (bf16*)save16->o : nullptr, out3_f8, nullptr, nullptr, nullptr);
    } else if (precision == 0) {
        if (out3) { set_error("attention forward: split output without the split kernel"); return -1; }
        const size_t lds = F_IMG + NPAD * HD * sizeof(float);
        bool* once = attr_flag(0);
        if (!*once) { if (set_lds((const void*)attn_fwd_f32_kernel, lds)) return -2; *once = true; }
        hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(grid), dim3(448), lds, s, (const float*)q, (const float*)k,
                           (const float*)v, (float*)out, lse);
    } else if (get_attn_v2() & 1) {
        return launch_attn_fwd_v2(q, k, v, out, lse, batch, s);
    } else {
        const size_t lds = ROW_IMG + TR_IMG;
        hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3(min(grid, 256)), dim3(448), lds, s, (const bf16*)q, (const bf16*)k,
                           (const bf16*)v, (bf16*)out, lse, grid);
    }
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

static int g_attn_v2 = -1;   // -1: not set yet -> DYT_ATTN_V2 from the environment (default 3: both round-5 kernels)
void set_attn_v2(int mask) { g_attn_v2 = mask; }
int get_attn_v2() {
    if (g_attn_v2 < 0) { const char* e = getenv("DYT_ATTN_V2"); g_attn_v2 = e ? (atoi(e) & 3) : 3; }
    return g_attn_v2;
}

static int g_attn_bwd_fused = 1;   // 16-bit modes: one kernel for dQ and dK/dV (0: the two separate kernels; 2: fused, row prefetch a head ahead)
void set_attn_bwd_fused(int on) { g_attn_bwd_fused = on; }

int launch_attn_bwd(int precision, const void* q, const void* k, const void* v, const void* out, const void* dout,
                    const float* lse, float* delta, void* dqkv, int batch, hipStream_t s, int q_tiles, void* dqkv3, float s3, int split16, int grad_parts, int out_hi_only,
                    int out_ld) {
    if (out_ld <= 0) out_ld = D;
    if (out_ld != D && precision == 0) { set_error("attention backward: a strided `out` is a 16-bit-kernel feature"); return -1; }
    const int grid = batch * NH;
    if (dbg_skip(1)) return 0;
    if (precision == 0 && (split16 || g_attn_f32_split)) {
        const size_t lds1 = 4 * ROW_H + 2 * TR_H, lds2 = 4 * ROW_H + 4 * TR_H + 2 * HROWS * sizeof(float);
        static bool done[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (!done[dev & 63]) {
            if (set_lds((const void*)attn_bwd_dq_split_kernel<3>, lds1) || set_lds((const void*)attn_bwd_dq_split_kernel<1>, lds1)) return -2;
            if (set_lds((const void*)attn_bwd_dkv_split_kernel<3>, lds2) || set_lds((const void*)attn_bwd_dkv_split_kernel<1>, lds2)) return -2;
            done[dev & 63] = true;
        }
        static const float gs_unit = getenv("DYT_SPLIT_ATTN_GS") ? (float)atof(getenv("DYT_SPLIT_ATTN_GS")) : 4096.0f;   // unit entries (no context)
        auto* kq = grad_parts >= 3 ? attn_bwd_dq_split_kernel<3> : attn_bwd_dq_split_kernel<1>;
        auto* kkv = grad_parts >= 3 ? attn_bwd_dkv_split_kernel<3> : attn_bwd_dkv_split_kernel<1>;
        hipLaunchKernelGGL(kq, dim3(min(grid, 256)), dim3(448), lds1, s, (const float*)q, (const float*)k, (const float*)v,
                           (const float*)out, (const float*)dout, lse, delta, (float*)dqkv, (bf16*)dqkv3, s3, grid, dqkv3 ? s3 : gs_unit, out_hi_only);
        hipLaunchKernelGGL(kkv, dim3(min(grid, 256)), dim3(448), lds2, s, (const float*)q, (const float*)k, (const float*)v,
                           (const float*)dout, lse, delta, (float*)dqkv, (bf16*)dqkv3, s3, grid, dqkv3 ? s3 : gs_unit, out_hi_only);
        DYT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (precision == 0) {
        const size_t lds1 = 2 * F_IMG;
        const size_t lds2 = 2 * F_IMG + 2 * NPAD * sizeof(float);
        bool* once = attr_flag(1);
        if (!*once) {
            if (set_lds((const void*)attn_bwd_dq_f32_kernel, lds1)) return -2;
            if (set_lds((const void*)attn_bwd_dkv_f32_kernel, lds2)) return -2;
            *once = true;
        }
        hipLaunchKernelGGL(attn_bwd_dq_f32_kernel, dim3(grid), dim3(448), lds1, s, (const float*)q, (const float*)k,
                           (const float*)v, (const float*)out, (const float*)dout, lse, delta, (float*)dqkv, (bf16*)dqkv3, s3);
        hipLaunchKernelGGL(attn_bwd_dkv_f32_kernel, dim3(grid), dim3(448), lds2, s, (const float*)q, (const float*)k,
                           (const float*)v, (const float*)dout, lse, delta, (float*)dqkv, (bf16*)dqkv3, s3);
    } else {
        const size_t lds1 = 2 * ROW_IMG + TR_IMG;
        const size_t lds2 = 2 * ROW_IMG + 2 * TR_IMG + 2 * NPAD * sizeof(float);
        bool* once = attr_flag(2);
        if (!*once) {
            if (set_lds((const void*)attn_bwd_dq_bf16_kernel, lds1)) return -2;
            if (set_lds((const void*)attn_bwd_dkv_bf16_kernel, lds2)) return -2;
            if (set_lds((const void*)attn_bwd_fused_bf16_kernel<false>, lds2)) return -2;
            if (set_lds((const void*)attn_bwd_fused_bf16_kernel<true>, lds2)) return -2;
            *once = true;
        }
        if (get_attn_v2() & 2) return launch_attn_bwd_v2(q, k, v, out, dout, lse, dqkv, batch, s, q_tiles, out_ld);
        if (g_attn_bwd_fused) {
            if (g_attn_bwd_fused == 2)
                hipLaunchKernelGGL(attn_bwd_fused_bf16_kernel<true>, dim3(min(grid, 256)), dim3(448), lds2, s, (const bf16*)q, (const bf16*)k,
                                   (const bf16*)v, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, grid, q_tiles, out_ld);
            else
                hipLaunchKernelGGL(attn_bwd_fused_bf16_kernel<false>, dim3(min(grid, 256)), dim3(448), lds2, s, (const bf16*)q, (const bf16*)k,
                                   (const bf16*)v, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, grid, q_tiles, out_ld);
            DYT_HIP_CHECK(hipGetLastError());
            return 0;
        }
        hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, dim3(min(grid, 256)), dim3(448), lds1, s, (const bf16*)q, (const bf16*)k,
                           (const bf16*)v, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, grid, out_ld);
        hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, dim3(min(grid, 256)), dim3(448), lds2, s, (const bf16*)q, (const bf16*)k,
                           (const bf16*)v, (const bf16*)dout, lse, delta, (bf16*)dqkv, grid);
    }
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dyt
