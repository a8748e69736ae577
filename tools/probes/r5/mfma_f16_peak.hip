// Probe: what v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 sustain on this box with NOTHING else running (operands in registers, no
// LDS, no memory), DVFS included: zero vs random operands (switching activity -> power -> clock), 1 or 2 waves per SIMD, 5-second runs so
// that the power controller has settled.  The "dense peak" of 2.5 PFLOP/s is quoted at 2.4 GHz; this is the ceiling a GEMM main loop
// can be held against on a 1.4 kW board.   hipcc --offload-arch=gfx950 -O3 -o /tmp/peak16 mfma_f16_peak.hip && /tmp/peak16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const _Float16* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = *(const h8*)(in + ((t * 64 + i * 8) & 0xFFFF8)); b[i] = *(const h8*)(in + ((t * 64 + 32 + i * 8) & 0xFFFF8)); }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + j) & 3], b[(i + 2 * j) & 3], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + j) & 3], b[(i + 2 * j) & 3], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    }
    out[t] = s;
}
template <int SHAPE> void run(const _Float16* in, float* out, const char* what, int wgs) {
    const int grid = 256 * wgs;
    // flops per iteration and wave: SHAPE 16: 32 MFMAs x 16*16*32*2; SHAPE 32: 16 MFMAs x 32*32*16*2  (both 524288)
    const double fl_iter = 524288.0 * 4 * grid;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE><<<grid, 256>>>(in, out, 1000);
    hipDeviceSynchronize();
    int iters = 200000;
    for (int rep = 0; rep < 2; ++rep) {   // second run: ~5 s
        hipEventRecord(e0);
        k<SHAPE><<<grid, 256>>>(in, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%dx%d f16, %s operands, %d wave(s)/SIMD: %.0f TFLOP/s over %.0f ms\n", SHAPE, SHAPE, what, wgs, fl_iter * iters / ms / 1e9, ms);
        fflush(stdout);
        iters = (int)(iters * 5000.0 / ms);
    }
}
int main() {
    _Float16* in; float* out;
    const int N = 1 << 20;
    hipMalloc(&in, N * 2); hipMalloc(&out, 256 * 8 * 256 * 4);
    _Float16* h = (_Float16*)malloc(N * 2);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < N; ++i) h[i] = mode ? (_Float16)((float)rand() / RAND_MAX - 0.5f) : (_Float16)0.f;
        hipMemcpy(in, h, N * 2, hipMemcpyHostToDevice);
        for (int wgs = 1; wgs <= 2; ++wgs) {
            run<16>(in, out, mode ? "random" : "zero", wgs);
            run<32>(in, out, mode ? "random" : "zero", wgs);
        }
    }
    return 0;
}
