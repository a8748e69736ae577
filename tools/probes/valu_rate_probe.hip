// Probe: issue cost (cycles per wave64 instruction) of the VALU ops the attention softmax is made of, one wave per SIMD and two,
// alone and next to MFMAs of the other wave of the SIMD.  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu valu_rate_probe.hip && /tmp/valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

template <int OP>
__global__ __launch_bounds__(64) void k(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* cyc) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 a8, b8; for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)v[i]; b8[i] = (__bf16)v[8 + i]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {          // 16 independent v_exp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        } else if (OP == 1) {   // 16 v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
        } else if (OP == 2) {   // 8 v_pk_fma_f32 (16 elements)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 r = __builtin_elementwise_fma(f32x2{v[i], v[i + 1]}, f32x2{0.999f, 0.999f}, f32x2{0.001f, 0.001f});
                v[i] = r[0]; v[i + 1] = r[1];
            }
        } else if (OP == 3) {   // 8 v_cvt_pk_bf16_f32 (+ 8 v_lshlrev to turn them back into floats)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const bf16x2 r = __builtin_convertvector(f32x2{v[i], v[i + 1]}, bf16x2);
                v[i] = (float)r[0]; v[i + 1] = (float)r[1];
            }
        } else if (OP == 4) {   // 4 MFMA 32x32x16 bf16 (dependent chain)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
        } else if (OP == 5) {   // 4 MFMA + 16 exp (one wave: can they overlap?)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        } else if (OP == 6) {   // even waves MFMA, odd waves exp (two waves of one SIMD at different work)
            if ((blockIdx.x >> 2) & 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, int per_iter, const float* in, float* out, unsigned long long* cyc) {
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int grid = 256 * 4 * waves_per_simd, iters = 20000;   // one 64-thread block = one wave; 4 SIMDs per CU
        k<OP><<<grid, 64>>>(in, out, 10, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<OP><<<grid, 64>>>(in, out, iters, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s %d wave/SIMD: %7.3f ms, %6.1f cycles (s_memtime 100 MHz units x?) per iteration per wave, %5.2f ns/iter -> %5.1f ns per instr group\n",
               name, waves_per_simd, ms, (double)c / iters, ms * 1e6 / iters, ms * 1e6 / iters / per_iter);
    }
}
int main() {
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 64 * 16 * 4); hipMalloc(&out, 256 * 8 * 64 * 4 * 2); hipMalloc(&cyc, 8);
    float h[64 * 16];
    for (int i = 0; i < 64 * 16; ++i) h[i] = -0.5f + (i % 7) * 0.1f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("16 x v_exp_f32", 16, in, out, cyc);
    run<1>("16 x v_fma_f32", 16, in, out, cyc);
    run<2>("8 x v_pk_fma_f32", 8, in, out, cyc);
    run<3>("8 x v_cvt_pk_bf16_f32 + 16 shifts", 24, in, out, cyc);
    run<4>("4 x mfma_32x32x16_bf16 (chain)", 4, in, out, cyc);
    run<5>("4 mfma + 16 exp, same wave", 20, in, out, cyc);
    run<6>("mfma waves next to exp waves", 1, in, out, cyc);
    return 0;
}
