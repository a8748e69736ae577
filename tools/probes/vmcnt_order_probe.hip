// Stand-alone hardware probe (gfx950): does `s_waitcnt vmcnt(1)` guarantee that the OLDER of two outstanding global loads has
// written its VGPR?  (That is the in-order assumption every partial vmcnt wait relies on.)
//
//   victim wave:  vA <- SENTINEL
//                 global_load_dword vA, cold[unique line]      (older load: misses L1 / L2)
//                 global_load_dword vB, hot[lane]              (younger load: a 256-byte table every wave reads -> L1 hit)
//                 s_waitcnt vmcnt(1)                           (in-order completion => vA has landed)
//                 vA == SENTINEL ?  -> count an error          (cold[] never contains the sentinel)
//
// run alone, and next to an aggressor stream whose workgroups (LDS transposes + bf16 MFMA + streaming global loads, two per CU, the
// shape of the adapter weight-gradient kernel) share the CUs.
//   hipcc --offload-arch=gfx950 -O2 -o vmcnt_order_probe vmcnt_order_probe.hip && ./vmcnt_order_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void victim(const unsigned* __restrict__ cold, const unsigned* __restrict__ hot, size_t stride_words,
                                              int iters, unsigned long long* __restrict__ errors, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned bad = 0, acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned* pc = cold + (wave * iters + it) * stride_words + lane;   // a fresh 256-byte line per wave and iteration
        const unsigned* ph = hot + lane;
        unsigned b;
        unsigned a2, snap;
        asm volatile(
            "v_mov_b32 %0, 0x5a5a5a5a\n\t"
            "s_nop 4\n\t"
            "global_load_dword %0, %2, off\n\t"
            "global_load_dword %3, %4, off\n\t"
            "s_waitcnt vmcnt(1)\n\t"
            "v_mov_b32 %1, %0\n\t"          /* vA as it is when the partial wait returns */
            "s_waitcnt vmcnt(0)"
            : "=&v"(a2), "=&v"(snap), "+v"(pc), "=&v"(b) : "v"(ph) : "memory");
        bad += (snap == 0x5a5a5a5au);
        acc += a2 + b + snap;
    }
    if (bad) atomicAdd(errors, (unsigned long long)bad);
    if (acc == 0x12345678u) sink[0] = acc;
}


typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// the load group of ln_bwd after its first drain, verbatim: LN weight (3 x dwordx4, scalar base + lane offset, the same 3 KB for every
// wave: L1 hits) with the row's (mean, rstd) pair (dwordx2, ONE address for all 64 lanes, 8 bytes per row: 16 rows share a line) issued
// second; the compiler waits vmcnt(2) before using the pair
__global__ __launch_bounds__(256) void victim_ln(const float* __restrict__ w, const float* __restrict__ stats, int rows,
                                                 unsigned long long* __restrict__ errors, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* ps = stats + (size_t)row * 2;
    unsigned voff = lane * 16;
    f32x4 w0, w1, w2;
    unsigned long long st = 0x5a5a5a5a5a5a5a5aull, snap;
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %5, %6\n\t"
        "global_load_dwordx2 %3, %4, off\n\t"
        "global_load_dwordx4 %1, %5, %6 offset:1024\n\t"
        "global_load_dwordx4 %2, %5, %6 offset:2048\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_mov_b64 %7, %3\n\t"           /* the (mean, rstd) pair as the wave would consume it */
        "s_waitcnt vmcnt(0)"
        : "=&v"(w0), "=&v"(w1), "=&v"(w2), "+v"(st), "+v"(ps), "+v"(voff), "+s"(w), "=&v"(snap) : : "memory");
    if (snap == 0x5a5a5a5a5a5a5a5aull) atomicAdd(errors, 1ull);
    if (w0[0] + w1[1] + w2[2] + (float)st == 1234.5f) sink[0] = 1.f;
}

// aggressor: LDS transposing stores + bf16 MFMA 32x32x16 + streaming loads, 27.6 KB LDS, <= 128 VGPRs: two workgroups per CU
__global__ __launch_bounds__(256, 2) void aggressor(const __bf16* __restrict__ X, float* __restrict__ out, int rows, int reps) {
    __shared__ __attribute__((aligned(16))) __bf16 Xt[128 * 72];
    __shared__ __attribute__((aligned(16))) __bf16 Yt[64 * 72];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int r = 0; r < reps; ++r) {
        const size_t row0 = ((size_t)blockIdx.x * reps + r) * 64 % (size_t)rows;
        bf16x8 x0 = *reinterpret_cast<const bf16x8*>(X + (row0 + (tid >> 4) * 2) * 768 + (tid & 15) * 8);
        bf16x8 x1 = *reinterpret_cast<const bf16x8*>(X + (row0 + (tid >> 4) * 2 + 1) * 768 + (tid & 15) * 8);
        __syncthreads();
        for (int i = 0; i < 8; ++i) {
            Xt[((tid & 15) * 8 + i) * 72 + 2 * (tid >> 4)] = x0[i];
            Xt[((tid & 15) * 8 + i) * 72 + 2 * (tid >> 4) + 1] = x1[i];
            if (tid < 128) { Yt[((tid & 7) * 8 + i) * 72 + 2 * (tid >> 3)] = x1[i]; Yt[((tid & 7) * 8 + i) * 72 + 2 * (tid >> 3) + 1] = x0[i]; }
        }
        __syncthreads();
        const int lr = lane & 31, lk = lane >> 5;
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&Xt[(wave * 32 + lr) * 72 + kk * 16 + lk * 8]);
            for (int j = 0; j < 2; ++j) {
                const bf16x8 yf = *reinterpret_cast<const bf16x8*>(&Yt[(j * 32 + lr) * 72 + kk * 16 + lk * 8]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf, acc[j], 0, 0, 0);
            }
        }
    }
    float t = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) t += acc[j][i];
    out[(size_t)blockIdx.x * 256 + tid] = t;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int vgrid = 6304, iters = 4;                       // the ln_bwd grid at B = 128
    const size_t stride_words = 64;                          // 256 B per (wave, iteration), two loads 128 B apart
    const size_t cold_words = (size_t)vgrid * 4 * iters * stride_words;
    unsigned *cold, *hot, *sink; unsigned long long* errors; __bf16* X; float* out;
    CK(hipMalloc(&cold, cold_words * 4 * 8));                // 8 rotating copies so that every launch reads lines that are not cached
    CK(hipMalloc(&hot, 256)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&errors, 8));
    const int rows = 25216;
    CK(hipMalloc(&X, (size_t)rows * 768 * 2)); CK(hipMalloc(&out, (size_t)512 * 256 * 4));
    CK(hipMemset(cold, 0x11, cold_words * 4 * 8)); CK(hipMemset(hot, 0x22, 256)); CK(hipMemset(X, 0x3c, (size_t)rows * 768 * 2));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float *lnw, *stats, *fsink;
    const int lrows = 25216;
    CK(hipMalloc(&lnw, 3072)); CK(hipMalloc(&stats, (size_t)lrows * 8 * 64)); CK(hipMalloc(&fsink, 4));
    CK(hipMemset(lnw, 0x3f, 3072)); CK(hipMemset(stats, 0x3e, (size_t)lrows * 8 * 64));
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(errors, 0, 8));
        for (int l = 0; l < launches * 4; ++l) {
            if (mode == 1 && (l & 3) == 0) hipLaunchKernelGGL(aggressor, dim3(480), dim3(256), 0, s2, X, out, rows, 40);
            hipLaunchKernelGGL(victim_ln, dim3((lrows + 3) / 4), dim3(256), 0, s1, lnw, stats + (size_t)(l % 64) * lrows * 2, lrows, errors, fsink);
        }
        CK(hipDeviceSynchronize());
        unsigned long long e = 0;
        CK(hipMemcpy(&e, errors, 8, hipMemcpyDeviceToHost));
        printf("ln_bwd load group %s: %llu of %llu (mean, rstd) pairs had NOT landed when s_waitcnt vmcnt(2) returned\n",
               mode == 0 ? "alone             " : "next to aggressor ", e, (unsigned long long)launches * 4 * lrows);
    }
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(errors, 0, 8));
        for (int l = 0; l < launches; ++l) {
            if (mode == 1) hipLaunchKernelGGL(aggressor, dim3(480), dim3(256), 0, s2, X, out, rows, 40);
            hipLaunchKernelGGL(victim, dim3(vgrid), dim3(256), 0, s1, cold + (size_t)(l % 8) * cold_words, hot, stride_words, iters, errors, sink);
        }
        CK(hipDeviceSynchronize());
        unsigned long long e = 0;
        CK(hipMemcpy(&e, errors, 8, hipMemcpyDeviceToHost));
        printf("%s: %llu of %llu older loads had NOT landed when s_waitcnt vmcnt(1) returned\n",
               mode == 0 ? "victim alone             " : "victim next to aggressor ", e, (unsigned long long)launches * vgrid * 256 * iters);
    }
    return 0;
}
