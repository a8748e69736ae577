// Round-5 attention kernels for the 16-bit modes (N = 197 tokens, 12 heads x 64): the same math as attention.hip
// (Attention.forward of the reference, models/vision_transformer_IN21K.py:60-70, and its autograd backward), restructured:
//   * ONE LDS image per operand.  K / V (forward) and Q / dO / K / V (backward) are row-major [224][64] 16-bit images filled by
//     LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass, no VALU); the 16-byte chunks of a row are XOR-swizzled
//     on the DMA *source* address (the DMA destination is lane-linear) so that both kinds of fragment reads are bank-conflict free:
//       row fragments        (MFMA operand = 8 consecutive channels of one row)   ds_read_b128
//       transposed fragments (MFMA operand = 4 + 4 consecutive ROWS of one channel) ds_read_b64_tr_b16 straight from the same image
//     -- the transposed LDS copies of attention.hip (and their 4-way conflicting ds_write_b32 scatter) are gone.
//   * forward: online softmax per 32-key tile (running max / sum, the rescale deferred until the max grows by more than 2^THR), so a
//     wave holds ONE score tile instead of seven: <= 128 VGPRs, 57 KB of LDS -> two 7-wave workgroups per CU whose load / MFMA / VALU
//     phases interleave (attention.hip: one workgroup per CU, every wave in the same phase).
#include "kernels.h"

namespace dyt {
namespace av2 {

constexpr int IROWS = 224;            // 7 tiles of 32 rows
constexpr int IMG = IROWS * 128;      // 28672 B per operand image
constexpr int PIECES = 25;            // DMA pieces of 8 rows that hold rows < 197

// chunk slot of row `row`: slot = chunk ^ swz(row).  (row >> 1) & 7 with its low bit moved to bit 2: the 16 lanes of a ds_read_b128
// service group (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a tile) then hit 16 different 16-byte slots of the 256-B bank row,
// and the four rows of a transposed read (r, r+1 | r+2, r+3) use complementary slot quadruples.
__device__ __forceinline__ int swz(int row) { const int x = (row >> 1) & 7; return ((x & 1) << 2) | (x >> 1); }

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB at the wave-uniform LDS byte address `lds`), as inline asm: hipcc tracks a
// __builtin_amdgcn_global_load_lds as a pending LDS write and puts a full `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot
// disambiguate (here: the first ds_read_b64_tr_b16 of the head being computed), which serialises the next head's DMA with this head's
// arithmetic.  Issued from asm the compiler does not see it; completion is awaited by the explicit vmcnt(0) + barrier at the top of a head.
// M0 (the DMA's LDS base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// rows [0, 197) of a [197][ld] 16-bit matrix (64 channels used) -> swizzled image; piece p = rows 8p .. 8p+7, pieces dealt to the waves
template <int NW>
__device__ __forceinline__ void dma_image(const bf16* __restrict__ src, int ld, unsigned img, int wave_s, int lane) {
    const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
        const int p = wave_s + i * NW;
        if (p < PIECES) {
            const int row = p * 8 + r8;
            if (row < NT)   // lanes of rows >= 197 stay inactive: the DMA leaves the (zeroed) pad rows alone
                dma16(src + (size_t)row * ld + ((slot ^ swz(row)) << 3), img + p * 1024);
        }
    }
}

// lane-constant parts of the fragment addresses (bytes inside an image)
struct FragAddr {
    int row[4];   // row fragment of tile row (lane & 31), channel chunk ks*2 + hi:  + tile * 4096
    int tr[2][2]; // transposed fragment [channel tile dt][second 4-row group x8]:    + tile * 4096 + half * 2048
    __device__ __forceinline__ FragAddr(int lane) {
        const int l31 = lane & 31, hi = lane >> 5, f = swz(l31);
        const int a0 = l31 * 128 + ((hi ^ (f & 1)) << 4) + ((f >> 1) << 5);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) row[ks] = a0 ^ (ks << 5);
        // ds_read_b64_tr_b16: inside a 16-lane group lane i supplies 4 consecutive channels of row (i >> 2) and receives channel i of
        // the 4 rows.  Group g16 = channels (g16 & 1) * 16 .. + 15 of the 32-channel tile, rows 4 * (g16 >> 1) .. + 3 of the 8-row group.
        const int g16 = lane >> 4, i = lane & 15;
        const int t0 = hi * 512 + (i >> 2) * 128 + (i & 1) * 8 + ((((i >> 1) & 1) ^ hi) << 4) + ((g16 & 1) << 5) + (((i >> 3) & 1) << 6);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int x8 = 0; x8 < 2; ++x8) tr[dt][x8] = (t0 ^ (dt << 6) ^ (x8 << 5)) + x8 * 1024;
    }
};

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int a_lo, int a_hi) {   // 4 rows at a_lo, 4 rows at a_hi (8 rows further)
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + a_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + a_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 row_frag(const char* img, int a) { return *reinterpret_cast<const bf16x8*>(img + a); }
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int base) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)v[base + i];
    return o;
}
// v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second: with two DISTINCT registers holding v the
// results are [v_lo | v_lo] and [v_hi | v_hi], i.e. every lane sees its own and its partner's (lane ^ 32) value.  (Handing the builtin the same
// SSA value twice lets the compiler use one register for both operands, which swaps that register's halves instead.)
__device__ __forceinline__ void xhalf_pair(float v, float& a, float& b) {
    unsigned x = __builtin_bit_cast(unsigned, v), y = x;
    asm("" : "+v"(y));
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    const unsigned r0 = r[0], r1 = r[1];   // scalars first: __builtin_bit_cast applied to the vector ELEMENT expression r[1] reads element 0 (clang 22)
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float xhalf_max(float v) { float a, b; xhalf_pair(v, a, b); return fmaxf(a, b); }   // over the two 32-lane halves, in every lane
__device__ __forceinline__ float xhalf_sum(float v) { float a, b; xhalf_pair(v, a, b); return a + b; }
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define MFMA32(a, b, c) DYT_MFMA_32x32x16((a), (b), (c))
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 8.0f * LOG2E;   // in log2 units: the running max is raised only when a tile exceeds it by more than e^8

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// Persistent: workgroup w walks the (image, head) pairs w, w + gridDim.x, ...  8 waves:
//   waves 0..6  one 32-row query tile each: S^T = K Q^T per 32-key tile, online softmax, O^T += V^T P^T.  They issue no loads at all
//               (operands come from the LDS images), so their stores stay in flight across heads;
//   wave 7      the loader: K / V of head i+1 into the other image pair, Q of head i+1 into the single Q image, by LDS-DMA, while head i
//               is computed (a DMA piece costs its issuing wave 60-185 cycles; 57 pieces per head would be ~1/4 of a compute wave's head).
// Two barriers per head: (a) head i's images have landed / every wave has left head i-1's K, V; (b) every wave holds its q fragments,
// the Q image may be overwritten.  PIPE: the score tile of key tile kt+1 is issued before the exponentials of tile kt.
// ABL (tools/probes/r5/av2_test.hip only): 1 = no output stores, 2 = no compute, 4 = no DMA after the first head
template <bool PIPE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attn_fwd_v2_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                             const bf16* __restrict__ v, bf16* __restrict__ out,
                                                             float* __restrict__ lse, int nheads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | V0 | K1 | V1 | Q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = wave * 32 + l31;
    for (int i = tid; i < 5 * IMG / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);   // pad rows stay zero for good
    __syncthreads();
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const char* Qi = smem + 4 * IMG;
    if (wave_s == 7) {   // ---- loader ----
        auto issue = [&](int bh, int buf) {
            dma_image<1>(q + (size_t)bh * NT * HD, HD, lds0 + 4 * IMG, 0, lane);
            dma_image<1>(k + (size_t)bh * NT * HD, HD, lds0 + buf * 2 * IMG, 0, lane);
            dma_image<1>(v + (size_t)bh * NT * HD, HD, lds0 + buf * 2 * IMG + IMG, 0, lane);
        };
        int bh = blockIdx.x, buf = 0;
        if (bh < nheads) issue(bh, 0);
        for (; bh < nheads; bh += gridDim.x, buf ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // (a)
            __builtin_amdgcn_s_barrier();   // (b)
            if (bh + gridDim.x < nheads && !(ABL & 4)) issue(bh + gridDim.x, buf ^ 1);
        }
        return;
    }
    const FragAddr fa(lane);
    int buf = 0;
    for (int bh = blockIdx.x; bh < nheads; bh += gridDim.x, buf ^= 1) {
        const int b = bh / NH, h = bh - b * NH;
        const char* Ki = smem + buf * 2 * IMG;
        const char* Vi = Ki + IMG;
        __builtin_amdgcn_s_barrier();   // (a)
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = row_frag(Qi, fa.row[ks] + wave * 4096);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // (b)
        float m = -INFINITY;   // in log2 units (scores times log2 e)
        f32x16 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
        auto kfrag = [&](int kt, bf16x8 (&kf)[4]) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = row_frag(Ki, fa.row[ks] + kt * 4096);
        };
        auto scores = [&](const bf16x8 (&kf)[4]) {   // S^T[key = kt*32 + (r&3) + 8*(r>>2) + 4*hi][q = l31]
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = MFMA32(kf[ks], qf[ks], s);
            return s;
        };
        bf16x8 kf[4];
        kfrag(0, kf);
        f32x16 s = scores(kf), sn;
        if (PIPE) kfrag(1, kf);
        float l4[4] = {0.f, 0.f, 0.f, 0.f};   // independent partial row sums (a single chain of 16 dependent adds per tile is latency-bound)
#pragma unroll
        for (int kt = 0; kt < ((ABL & 2) ? 0 : 7); ++kt) {
            const int nh = kt < 6 ? 2 : 1;   // keys 208 .. 223 are all padding
            bf16x8 vf[2][2];
#pragma unroll
            for (int half = 0; half < nh; ++half)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    vf[half][dt] = tr_frag(Vi, fa.tr[dt][0] + kt * 4096 + half * 2048, fa.tr[dt][1] + kt * 4096 + half * 2048);
            if (kt == 6) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (192 + (r & 3) + 8 * (r >> 2) + 4 * hi >= NT) s[r] = -INFINITY;
            }
            // tile maximum: v_max3 straight on the accumulators (fmaxf() puts a canonicalising v_max in front of every MFMA output)
            float mt = max3(max3(s[0], s[1], s[2]), max3(s[3], s[4], s[5]), max3(s[6], s[7], s[0]));
            if (kt < 6) mt = max3(mt, max3(max3(s[8], s[9], s[10]), max3(s[11], s[12], s[13]), max3(s[14], s[15], s[8])), mt);
            mt = xhalf_max(mt) * LOG2E;
            if (__any(mt > m + RESCALE_THR)) {   // wave-uniform: raise the running max of every row of the tile, rescale l and O
                const float mn = fmaxf(m, mt);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
#pragma unroll
                for (int i = 0; i < 4; ++i) l4[i] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            }
            // from here on one basic block: the next tile's score MFMAs (independent of everything below) are issued first and the
            // exponentials of this tile run under them; then P V
            if (PIPE && kt < 6) {
                sn = scores(kf);
                if (kt < 5) kfrag(kt + 2, kf);
            }
#pragma unroll
            for (int half = 0; half < nh; ++half) {
#pragma unroll
                for (int r = half * 8; r < half * 8 + 8; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -m));
                    s[r] = p;
                    l4[r & 3] += p;
                }
                const bf16x8 pf = pack8(s, half * 8);
                o[0] = MFMA32(vf[half][0], pf, o[0]);   // O^T[d][q] += V^T[d][key] P^T[key][q]
                o[1] = MFMA32(vf[half][1], pf, o[1]);
            }
            if (kt < 6) {
                if (PIPE) s = sn;
                else { kfrag(kt + 1, kf); s = scores(kf); }
            }
        }
        const float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
        const float sum = xhalf_sum(l);
        if (ABL & 1) { if (sum == 123.456f) lse[0] = o[0][0] + o[1][5] + s[3]; continue; }
        if (hi == 0 && qrow < NT) lse[(size_t)bh * NT + qrow] = (m + __builtin_amdgcn_logf(sum)) * (1.0f / LOG2E);
        // row q of O^T: this lane holds channels dt*32 + 8g + 4hi + 0..3 (g = 0..3); v_permlane32_swap pairs the groups (g, g+1) of the two
        // half-waves so that every lane stores 16 contiguous bytes (T21 of the CDNA4 guide): 4 stores of 16 B instead of 8 of 8 B
        const float inv = 1.0f / sum;
        bf16* op = out + ((size_t)b * NT + min(qrow, NT - 1)) * D + h * HD + hi * 8;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const bf16x4 a4 = {(bf16)(o[dt][4 * g] * inv), (bf16)(o[dt][4 * g + 1] * inv), (bf16)(o[dt][4 * g + 2] * inv), (bf16)(o[dt][4 * g + 3] * inv)};
                const bf16x4 b4 = {(bf16)(o[dt][4 * g + 4] * inv), (bf16)(o[dt][4 * g + 5] * inv), (bf16)(o[dt][4 * g + 6] * inv), (bf16)(o[dt][4 * g + 7] * inv)};
                const uint2 a = __builtin_bit_cast(uint2, a4), bb = __builtin_bit_cast(uint2, b4);
                const auto rx = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
                const unsigned rx0 = rx[0], rx1 = rx[1], ry0 = ry[0], ry1 = ry[1];
                if (qrow < NT) *reinterpret_cast<uint4*>(op + dt * 32 + 8 * g) = make_uint4(rx0, ry0, rx1, ry1);
            }
    }
}

}  // namespace av2

static int set_lds_v2(const void* fn, size_t bytes) {
    DYT_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

int launch_attn_fwd_v2(const void* q, const void* k, const void* v, void* out, float* lse, int batch, hipStream_t s) {
    const int grid = batch * NH;
    const size_t lds = 5 * av2::IMG;
    static bool done[64] = {};
    static const int pipe = getenv("DYT_ATTN_V2_PIPE") ? atoi(getenv("DYT_ATTN_V2_PIPE")) : 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (!done[dev & 63]) {
        if (set_lds_v2((const void*)av2::attn_fwd_v2_kernel<true>, lds) || set_lds_v2((const void*)av2::attn_fwd_v2_kernel<false>, lds)) return -2;
        done[dev & 63] = true;
    }
    auto* kern = pipe ? av2::attn_fwd_v2_kernel<true> : av2::attn_fwd_v2_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(min(grid, 256)), dim3(512), lds, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, lse, grid);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dyt
