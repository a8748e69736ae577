// Split-K form of the skinny GEMMs of the cls-only last block (included by gemm.hip).
//
// Only the cls row of the last block's output reaches the head, so that block's MLP runs on M = batch rows (DYT_OPT_CLS_TAIL).  Its two
// K = 3072 contractions -- fc2 forward (with the adapter's up-projection as a second operand pair) and the fc1 dgrad -- were 6 tiles of
// 128x128 on 256 CUs: 48 k-steps of a two-stage ring with four waves on the CU, 48-56 us each, at the point of the step where both
// passes are in their tails and nothing else is there to run beside them.  Here a workgroup takes a 128-row x 32-column block of ONE
// 256-wide k slice: grid = (N / 32) x ceil(M / 128) x (K / 256 [+ 1 for the second pair]) = 312 workgroups at B = 128, no LDS: a lane's
// MFMA fragment rows are 64 contiguous bytes of its A row / W row, so it loads them straight into registers (all 32 loads of the slice in
// flight at once).  The fp32 partial blocks go to a workspace the caller lends ([slices][M][N]); a second launch sums the slices in a
// FIXED order and runs the GEMM's epilogue functor on the sums (same col_init / pre / apply interface as the tile kernels' staged epilogue),
// so the result is deterministic, and equal to the tile kernels' up to the order of the fp32 additions over k.
#pragma once

namespace dyt {

constexpr int SK_SLICE = 256;   // k per workgroup: 16 MFMA steps of 32x32x16
constexpr int SK_MAX_M = 512;   // above this the 128x128 tiles fill enough CUs and the partials' round trip costs more than it saves

// k assignment inside a 64-wide block: lane half h = lane >> 5 takes k in [32 h, 32 h + 32), MFMA step s its chunk [8 s, 8 s + 8) -- the
// same on both operands, so any assignment is a valid contraction order; this one makes a lane's four fragments of a block one 64-B run.
template <bool CAT>
__global__ __launch_bounds__(256) void gemm_splitk_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K,
                                                          const int* __restrict__ a_map, const bf16* __restrict__ A2,
                                                          const bf16* __restrict__ W2, const int* __restrict__ a2_map,
                                                          float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
    const int row0 = blockIdx.y * 128 + wave * 32, n = blockIdx.x * 32 + l31, sl = blockIdx.z;
    if (row0 >= M) return;
    const int row = min(row0 + l31, M - 1);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nsl = K / SK_SLICE;
    if (CAT && sl == nsl) {   // the second operand pair: A2 [rows, 64] x W2 [N, 64]
        const int grow = a2_map ? a2_map[row] : row;
        const bf16x8* ap = reinterpret_cast<const bf16x8*>(A2 + (size_t)grow * 64 + h * 32);
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(W2 + (size_t)n * 64 + h * 32);
        bf16x8 fa[4], fw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { fa[s] = ap[s]; fw[s] = wp[s]; }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = DYT_MFMA_32x32x16(fa[s], fw[s], acc);
    } else {
        const int grow = a_map ? a_map[row] : row;
        const bf16x8* ap = reinterpret_cast<const bf16x8*>(A + (size_t)grow * K + sl * SK_SLICE + h * 32);
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(W + (size_t)n * K + sl * SK_SLICE + h * 32);
        bf16x8 fa[16], fw[16];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) { fa[b * 4 + s] = ap[b * 8 + s]; fw[b * 4 + s] = wp[b * 8 + s]; }   // block b = 64 k = 8 chunks of 16 B
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = DYT_MFMA_32x32x16(fa[t], fw[t], acc);
    }
    // D layout of the 32x32 MFMA: lane = column l31, register i = row 8 (i / 4) + 4 h + i % 4
    float* p = part + ((size_t)sl * M + row0) * N + n;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = 8 * (i >> 2) + 4 * h + (i & 3);
        if (row0 + r < M) p[(size_t)r * N] = acc[i];
    }
}

// sums of the slices in slice order + the epilogue functor; one thread = 4 consecutive columns of a row (16 consecutive lanes = 64
// consecutive columns of one row, the geometry the functors' row-group reductions assume: N / 4 is a multiple of 16)
template <class Epi>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int slices, int M, int N, float out_scale, Epi epi) {
    const int idx = blockIdx.x * 256 + threadIdx.x, nc = N >> 2;
    const int row = idx / nc, col = (idx - row * nc) * 4;
    if (row >= M) return;
    const typename Epi::Col c = epi.col_init(col);
    const typename Epi::Pre p = epi.pre(row, col);
    const float* src = part + (size_t)row * N + col;
    const size_t stride = (size_t)M * N;
    f32x4 s = *reinterpret_cast<const f32x4*>(src);
    for (int k = 1; k < slices; ++k) s += *reinterpret_cast<const f32x4*>(src + k * stride);
    const float a[4] = {s[0] * out_scale, s[1] * out_scale, s[2] * out_scale, s[3] * out_scale};
    epi.apply(row, col, a, c, p);
}

// epilogues that have this form (the two K = 3072 GEMMs of the cls tail, 16-bit modes)
template <class E> struct SplitKEpi : std::false_type {};
template <class HT> struct SplitKEpi<EpiFc2<bf16, false, HT>> : std::true_type {};
template <> struct SplitKEpi<EpiStoreAT<bf16>> : std::true_type {};

}  // namespace dyt
