// Probe: does a VALU row kernel (LayerNorm-backward-like: fp32 loads, packed-fp32 math, DPP wave reductions) give
// bit-identical results when a small bf16-MFMA kernel runs beside it on another stream?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/vmp valu_mfma_probe.hip && /tmp/vmp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_f32(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
// one wave per 768-channel row
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ __launch_bounds__(256) void row_kernel(const __bf16* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ base, float* __restrict__ out, __bf16* __restrict__ out16, int rows) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float g[12], xv[12], wv[12], b[12];
    for (int i = 0; i < 3; ++i) {
        const bf16x4 a16 = *reinterpret_cast<const bf16x4*>(dy + (size_t)row * 768 + i * 256 + lane * 4);
        const f32x4 a = {(float)a16[0], (float)a16[1], (float)a16[2], (float)a16[3]};
        const f32x4 c = *reinterpret_cast<const f32x4*>(x + (size_t)row * 768 + i * 256 + lane * 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(w + i * 256 + lane * 4);
        const f32x4 e = *reinterpret_cast<const f32x4*>(base + (size_t)row * 768 + i * 256 + lane * 4);
        for (int k = 0; k < 4; ++k) { g[4 * i + k] = a[k]; xv[4 * i + k] = c[k]; wv[4 * i + k] = d[k]; b[4 * i + k] = e[k]; }
    }
    float m = 0.f;
    for (int i = 0; i < 12; ++i) m += xv[i];
    m = wave_sum(m) * (1.0f / 768);
    float q = 0.f;
    for (int i = 0; i < 12; ++i) { const float d = xv[i] - m; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 768) + 1e-6f);
    float s1 = 0.f, s2 = 0.f, xh[12];
    for (int i = 0; i < 12; ++i) { xh[i] = (xv[i] - m) * rstd; g[i] *= wv[i]; s1 += g[i]; s2 = fmaf(g[i], xh[i], s2); }
    s1 = wave_sum(s1) * (1.0f / 768);
    s2 = wave_sum(s2) * (1.0f / 768);
    for (int i = 0; i < 3; ++i) {
        f32x4 o;
        for (int k = 0; k < 4; ++k) o[k] = b[4 * i + k] + rstd * (g[4 * i + k] - s1 - xh[4 * i + k] * s2);
        *reinterpret_cast<f32x4*>(out + (size_t)row * 768 + i * 256 + lane * 4) = o;
        *reinterpret_cast<bf16x4*>(out16 + (size_t)row * 768 + i * 256 + lane * 4) = bf16x4{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
    }
}
// small MFMA kernel: few workgroups, 4 waves each, bursts of v_mfma_f32_16x16x32_bf16 between barriers
__global__ __launch_bounds__(256) void mfma_kernel(float* __restrict__ out, int iters, int use_mfma) {
    __shared__ __attribute__((aligned(16))) __bf16 sm[128 * 72];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 72; i += 256) sm[i] = (__bf16)(0.001f * (i & 63));
    __syncthreads();
    f32x4 acc[10];
    for (int j = 0; j < 10; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        bf16x8 xf[2], yf[4];
        for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(&sm[(wave * 32 + i * 16 + (lane & 15)) * 72 + (lane >> 4) * 8]);
        for (int j = 0; j < 4; ++j) yf[j] = *reinterpret_cast<const bf16x8*>(&sm[(j * 16 + (lane & 15)) * 72 + 32 + (lane >> 4) * 8]);
        if (use_mfma) {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 4; ++j) acc[i * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf[j], acc[i * 5 + j], 0, 0, 0);
        } else {
            for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(xf[i]));
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(yf[j]));
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int j = 0; j < 10; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + tid] = s;
}
int main() {
    const int rows = 788, n = rows * 768;
    float *dy32, *x, *w, *base, *out, *ref, *mo; __bf16 *dy, *out16;
    hipMalloc(&dy32, n * 4); hipMalloc(&dy, n * 2); hipMalloc(&out16, n * 2); hipMalloc(&x, n * 4); hipMalloc(&w, 768 * 4); hipMalloc(&base, n * 4); hipMalloc(&out, 2 * n * 4); hipMalloc(&ref, n * 4);
    hipMalloc(&mo, 1024 * 256 * 4);
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 1e-3f;
    { unsigned short* hb = (unsigned short*)malloc(n * 2); for (int i = 0; i < n; ++i) { unsigned u; memcpy(&u, &h[i], 4); hb[i] = (unsigned short)(u >> 16); } hipMemcpy(dy, hb, n * 2, hipMemcpyHostToDevice); }
    for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 4.f;
    hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 1e-3f;
    hipMemcpy(base, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 768; ++i) h[i] = 1.0f + 0.1f * ((float)rand() / RAND_MAX - 0.5f);
    hipMemcpy(w, h, 768 * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    row_kernel<<<(rows + 3) / 4, 256, 0, s1>>>(dy, x, w, base, ref, out16, rows);
    hipDeviceSynchronize();
    float* href = (float*)malloc(n * 4); float* hout = (float*)malloc(n * 4);
    hipMemcpy(href, ref, n * 4, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 3; ++mode) {   // 0: alone, 1: beside the LDS-only kernel, 2: beside the MFMA kernel
        int bad_runs = 0, bad_rows = 0;
        for (int rep = 0; rep < 200; ++rep) {
            // a long-running neighbour on every CU (1024 workgroups x 4 waves, ~ms), then 20 row-kernel launches under it
            if (mode) mfma_kernel<<<1024, 256, 0, s2>>>(mo, 4000, mode == 2);
            for (int k = 0; k < 20; ++k) {
                row_kernel<<<(rows + 3) / 4, 256, 0, s1>>>(dy, x, w, base, out + (size_t)(k % 2) * n, out16, rows);
            }
            hipDeviceSynchronize();
            for (int k = 0; k < 2; ++k) {
                hipMemcpy(hout, out + (size_t)k * n, n * 4, hipMemcpyDeviceToHost);
                if (memcmp(hout, href, n * 4)) {
                    ++bad_runs;
                    for (int r = 0; r < rows; ++r) bad_rows += memcmp(hout + r * 768, href + r * 768, 768 * 4) != 0;
                }
            }
        }
        printf("mode %d: %d of 400 checked results differ from the reference (%d rows in total)\n", mode, bad_runs, bad_rows);
    }
    return 0;
}
