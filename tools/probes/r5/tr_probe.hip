// ds_read_b64_tr_b16 mapping probe (gfx950): which LDS halfword does lane l receive as element j?
// LDS holds halfword index i at position i (as uint16).  Each lane supplies byte address addr[l] (host-chosen).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__global__ void probe(const int* __restrict__ addr, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int a = addr[threadIdx.x];
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(reinterpret_cast<__attribute__((address_space(3))) char*>(
        (__attribute__((address_space(3))) void*)lds) + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
static void run(const char* name, std::vector<int> a) {
    int* da; uint16_t* dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&dout, 64 * 4 * 2);
    hipMemcpy(da, a.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dout);
    std::vector<uint16_t> o(256);
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("== %s\n", name);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d addr %4d -> %4d %4d %4d %4d", l, a[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
        // decode: source lane / element assuming lane-linear image addr = lane*8
        printf("\n");
    }
    hipFree(da); hipFree(dout);
}
int main() {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) a[l] = l * 8;           // lane-linear: lane l owns halfwords 4l..4l+3
    run("linear (addr = lane*8)", a);
    // a [4 rows][16 cols] block per 16-lane group with row stride 128 B: lane i -> row i/4, col 4*(i%4)
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; a[l] = g * 1024 + (i >> 2) * 128 + (i & 3) * 8; }
    run("4x16 blocks, row stride 128 B, group stride 1024 B", a);
    return 0;
}
