// Probe (MI355X): operand layout, scale semantics and issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands, and the
// behaviour of v_cvt_pk_fp8_f32 at the edges -- what the fp8-correction phase of the split GEMM (csrc/gemm.hip) relies on.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mxp tools/probes/mx_f8_probe.hip && /tmp/mxp
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void cvt_kernel(const float* x, unsigned char* o, int n) {   // 4 floats -> 4 fp8 bytes per thread
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 >= n) return;
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[i * 4], x[i * 4 + 1], v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[i * 4 + 2], x[i * 4 + 3], v, true);
    reinterpret_cast<int*>(o)[i] = v;
}
// A8 [16][128], B8 [16][128] bytes; mode 0: lane (r, g) takes bytes [32g, 32g+32); mode 1: chunks {g, 4+g} (16 B each)
__global__ void mfma_kernel(const unsigned char* A8, const unsigned char* B8, float* C, int sa, int sb, int mode) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v4i a0, a1, b0, b1;
    if (mode == 0) {
        a0 = *reinterpret_cast<const v4i*>(A8 + r * 128 + 32 * g); a1 = *reinterpret_cast<const v4i*>(A8 + r * 128 + 32 * g + 16);
        b0 = *reinterpret_cast<const v4i*>(B8 + r * 128 + 32 * g); b1 = *reinterpret_cast<const v4i*>(B8 + r * 128 + 32 * g + 16);
    } else {
        a0 = *reinterpret_cast<const v4i*>(A8 + r * 128 + 16 * g); a1 = *reinterpret_cast<const v4i*>(A8 + r * 128 + 64 + 16 * g);
        b0 = *reinterpret_cast<const v4i*>(B8 + r * 128 + 16 * g); b1 = *reinterpret_cast<const v4i*>(B8 + r * 128 + 64 + 16 * g);
    }
    const v8i A = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), B = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, 0, sa, 0, sb);
    for (int e = 0; e < 4; ++e) C[l * 4 + e] = acc[e];
}
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned long long* cyc) {
    v8i A, B; f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { A[i] = 0x38383838 + threadIdx.x; B[i] = 0x30303030 + i; ah[i] = (_Float16)(0.01f * i); bh[i] = (_Float16)(0.02f * threadIdx.x); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[i], 0, 0, 0, 127, 0, 127);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
static float dec_e4m3(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9); else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
int main() {
    // ---- conversion edges
    const float edge[16] = {0.f, 1.f, -1.f, 448.f, 464.f, 480.f, 1000.f, -1e6f, 0.0019f, 0.00098f, 0.0156f, 1.0625f, 1.1875f, INFINITY, NAN, 3.1416f};
    float* dx; unsigned char* d8; unsigned char h8[16];
    hipMalloc(&dx, sizeof(edge)); hipMalloc(&d8, 16);
    hipMemcpy(dx, edge, sizeof(edge), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dx, d8, 16);
    hipMemcpy(h8, d8, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", edge[i], h8[i], dec_e4m3(h8[i]));
    // ---- layout
    srand(1);
    float fa[16 * 128], fb[16 * 128];
    for (int i = 0; i < 16 * 128; ++i) { fa[i] = (rand() % 31 - 15) / 8.0f; fb[i] = (rand() % 31 - 15) / 4.0f; }
    float *dfa, *dfb, *dC; unsigned char *dA8, *dB8;
    hipMalloc(&dfa, sizeof(fa)); hipMalloc(&dfb, sizeof(fb)); hipMalloc(&dA8, 2048); hipMalloc(&dB8, 2048); hipMalloc(&dC, 1024);
    hipMemcpy(dfa, fa, sizeof(fa), hipMemcpyHostToDevice); hipMemcpy(dfb, fb, sizeof(fb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_kernel, dim3(2), dim3(256), 0, 0, dfa, dA8, 2048);
    hipLaunchKernelGGL(cvt_kernel, dim3(2), dim3(256), 0, 0, dfb, dB8, 2048);
    unsigned char hA[2048];
    hipMemcpy(hA, dA8, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2048; ++i) bad += dec_e4m3(hA[i]) != fa[i];
    printf("conversion of n/8 values exact: %s\n", bad ? "NO" : "yes");
    double ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 128; ++k) s += (double)fa[i * 128 + k] * fb[j * 128 + k]; ref[i][j] = s; }
    for (int mode = 0; mode < 2; ++mode)
        for (int sc = 0; sc < 3; ++sc) {
            const int sa = sc == 0 ? 127 : (sc == 1 ? 120 : 127), sb = sc == 2 ? 130 : 127;
            float hC[256];
            hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA8, dB8, dC, sa, sb, mode);
            hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
            const double f = ldexp(1.0, (sa - 127) + (sb - 127));
            double e1 = 0, e2 = 0;   // hypothesis 1: D[row = A row = (l>>4)*4 + e][col = B row = l&15]; hypothesis 2: transposed
            for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
                const int i = (l >> 4) * 4 + e, j = l & 15;
                e1 = fmax(e1, fabs(hC[l * 4 + e] - f * ref[i][j])); e2 = fmax(e2, fabs(hC[l * 4 + e] - f * ref[j][i]));
            }
            printf("layout mode %d scale_a %d scale_b %d: max|err| D[Arow=(l>>4)*4+e][Brow=l&15] %.3g, transposed %.3g\n", mode, sa, sb, e1, e2);
        }
    // ---- rate
    float* dout; unsigned long long* dcyc; unsigned long long cyc;
    hipMalloc(&dout, 256 * 1024 * 4); hipMalloc(&dcyc, 8);
    for (int kind = 0; kind < 2; ++kind)
        for (int waves = 1; waves <= 2; ++waves) {   // 1 or 2 waves per SIMD (256 / 512 threads... two blocks of 256 on one CU would need co-residency: use grid 256 blocks vs 512)
            const int iters = 4000;
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256 * waves), dim3(256), 0, 0, dout, iters, dcyc);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(256 * waves), dim3(256), 0, 0, dout, iters, dcyc);
            hipDeviceSynchronize();
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256 * waves), dim3(256), 0, 0, dout, iters, dcyc);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(256 * waves), dim3(256), 0, 0, dout, iters, dcyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
            const double flop = (double)256 * waves * 4 * iters * 8 * (kind == 0 ? 2.0 * 16 * 16 * 128 : 2.0 * 16 * 16 * 32);
            printf("%s, %d block(s) per CU: %.1f cycles per MFMA per wave (8 independent accumulators), %.0f TFLOP/s\n",
                   kind == 0 ? "mfma_scale 16x16x128 fp8" : "mfma 16x16x32 f16", waves, (double)cyc / (iters * 8.0), flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
