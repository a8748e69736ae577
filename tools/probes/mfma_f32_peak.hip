// Probe: ceiling of v_mfma_f32_32x32x2_f32 on this box (DVFS included) with random vs zero operands,
// 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o /tmp/peak mfma_f32_peak.hip && /tmp/peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xFFFFF]; b[i] = in[(t * 16 + 8 + i) & 0xFFFFF]; }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) & 7], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[i], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 3) & 7], b[(i + 5) & 7], acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[t] = s;
}
int main() {
    float *in, *out;
    const int N = 1 << 20;
    hipMalloc(&in, N * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    float* h = (float*)malloc(N * 4);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < N; ++i) h[i] = mode ? (float)rand() / RAND_MAX - 0.5f : 0.f;
        hipMemcpy(in, h, N * 4, hipMemcpyHostToDevice);
        for (int wgs = 1; wgs <= 2; ++wgs) {
            const int iters = 20000, grid = 256 * wgs;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<grid, 256>>>(in, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<grid, 256>>>(in, out, iters);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)grid * 4 * iters * 32 * 4096.0;
            printf("%s operands, %d waves/SIMD: %.1f TFLOP/s (%.2f ms)\n", mode ? "random" : "zero", wgs, fl / ms / 1e9, ms);
        }
    }
    return 0;
}
