// Wave-per-row helpers shared by the row-wise kernels (rowops.hip, pool.hip): one 64-lane wave owns one
// 768-channel token row (3 x float4 per lane, fully coalesced); reductions are wave shuffles.
#pragma once
#include "dyt_common.h"

namespace dyt {

// a wave's view of one 768-float row: lane holds cols {lane*4 + 256*i + e}
struct Row12 {
    float v[12];
    __device__ __forceinline__ void load(const float* p, int lane) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(p + i * 256 + lane * 4);
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    }
    // streaming variant: saved activations that the backward pass reads exactly once (keeps them out of the L2 lines
    // the concurrently running GEMMs are re-using)
    __device__ __forceinline__ void load_nt(const float* p, int lane) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 t = DYT_NT_LOAD(reinterpret_cast<const f32x4*>(p + i * 256 + lane * 4));
            v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
        }
    }
    template <class T>
    __device__ __forceinline__ void load_at(const T* p, int lane) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float t[4];
            load4(p + i * 256 + lane * 4, t);
            v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
        }
    }
    // the row times `scale` as the 16-bit hi / lo split operand of a GEMM (row base p3 of a [rows][SPLIT_A * 768] image)
    __device__ __forceinline__ void store_split3(bf16* p3, int lane, float scale, bool hi_only = false) const {   // hi_only: the consumer contracts the hi part alone
#pragma unroll
        for (int i = 0; i < 3; ++i)
            store4_split3(p3 + i * 256 + lane * 4, 768, v[4 * i] * scale, v[4 * i + 1] * scale, v[4 * i + 2] * scale, v[4 * i + 3] * scale, hi_only);
    }
    template <class T>
    __device__ __forceinline__ void store(T* p, int lane) const {
#pragma unroll
        for (int i = 0; i < 3; ++i) store4(p + i * 256 + lane * 4, v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    // full vmcnt drain ordered after the three load instructions that filled this row (see DYT_PIN* in dyt_common.h)
    __device__ __forceinline__ void landed() { DYT_PIN3(v[0], v[4], v[8]); }
    __device__ __forceinline__ float sum() const {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) s += v[i];
        return wave_sum(s);
    }
};

__device__ __forceinline__ float dot12(const Row12& a, const Row12& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s = fmaf(a.v[i], b.v[i], s);
    return wave_sum(s);
}

// normalise a row held in registers; returns (mean, rstd)
__device__ __forceinline__ float2 ln_stats(const Row12& x) {
    const float mean = x.sum() * (1.0f / D);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float d = x.v[i] - mean; s = fmaf(d, d, s); }
    const float var = wave_sum(s) * (1.0f / D);
    return make_float2(mean, 1.0f / sqrtf(var + LN_EPS));
}

__device__ __forceinline__ void ln_bwd_row(Row12& dy, const Row12& x, const Row12& w, float2 st) {
    // in: dy = dL/d(LN out); out: dy = dL/dx
    float s1 = 0.f, s2 = 0.f;
    Row12 xh;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        xh.v[i] = (x.v[i] - st.x) * st.y;
        dy.v[i] *= w.v[i];
        s1 += dy.v[i];
        s2 = fmaf(dy.v[i], xh.v[i], s2);
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < 12; ++i) dy.v[i] = st.y * (dy.v[i] - s1 - xh.v[i] * s2);
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace dyt
