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
// Maximum of the first 8 / all 16 registers of an MFMA accumulator, v_max3 straight on the registers (fmaxf() puts a canonicalising v_max
// in front of every MFMA output: 16 more VALU instructions per tile).  ONE asm statement that opens with the wait states the
// matrix-core -> VALU read hazard needs (8-pass MFMA: 12): hipcc pads hazards for its own instructions only, and a bare `v_max3` on the
// accumulators read them while the last MFMA of the chain was still writing whenever the SIMD's other wave kept the matrix pipe busy --
// slightly different running maxima, i.e. output bits that changed from run to run (the wave that shares its SIMD with the loader never did).
__device__ __forceinline__ float acc_max8(const f32x16& s) {
    float t0, t1;
    asm volatile("s_nop 11\n\tv_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %5, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\tv_max_f32 %0, %0, %1"
                 : "=&v"(t0), "=&v"(t1)
                 : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]));
    return t0;
}
__device__ __forceinline__ float acc_max16(const f32x16& s) {
    float t0, t1;
    asm volatile("s_nop 11\n\tv_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %5, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\tv_max3_f32 %1, %1, %10, %11\n\t"
                 "v_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %1, %1, %14, %15\n\tv_max3_f32 %0, %0, %16, %17\n\tv_max_f32 %0, %0, %1"
                 : "=&v"(t0), "=&v"(t1)
                 : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]), "v"(s[10]),
                   "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]));
    return t0;
}
#define MFMA32(a, b, c) DYT_MFMA_32x32x16((a), (b), (c))
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 8.0f * LOG2E;   // in log2 units: the running max is raised only when a tile exceeds it by more than e^8

// Result tiles (X^T accumulators: lane = row of X, registers = channels) -> global rows, in two steps so that the stores can be issued
// one at a time from inside the NEXT tile loop (a CU drains stores at ~7 B/clk; a burst of 24 stores stalls the wave for microseconds).
// pack_rows: 16-bit conversion, then three lane exchanges that turn "32 bytes of 32 rows per store instruction" into "all 128 bytes of 8
// rows" (the store path's cost is per cache line touched: 110 -> 102 us with 64-byte segments, -> with whole rows):
//   v_permlane32_swap  the two half-waves' 4-channel groups        -> 8 channels = 16 B per lane
//   v_permlane16_swap  the two 16-byte chunk pairs of a row        -> lanes r, 16 + r, 32 + r, 48 + r hold four adjacent chunks of row r (16 rows)
//   v_mov_dpp row_ror:8 with a bank mask: the two channel tiles    -> lanes r, 8 + r, ... hold the eight chunks of row r (8 rows)
// pk[p], p = half * 2 + sub: tile rows half*16 + sub*8 + (lane & 7); this lane holds the chunk 4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + (lane >> 5).
template <int BANKS> __device__ __forceinline__ unsigned dpp_ror8(unsigned old, unsigned src) {   // lanes of the banks in BANKS: src of lane ^ 8; others: old
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x128, 0xF, BANKS, false);
}
__device__ __forceinline__ void pack_rows(uint4 (&pk)[4], const f32x16 (&acc)[2], float scale) {
    unsigned t[2][2][4];   // [dt][half][dword]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        unsigned w[2][4];
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const int g = 2 * gp;
            const bf16x4 a4 = {(bf16)(acc[dt][4 * g] * scale), (bf16)(acc[dt][4 * g + 1] * scale), (bf16)(acc[dt][4 * g + 2] * scale), (bf16)(acc[dt][4 * g + 3] * scale)};
            const bf16x4 b4 = {(bf16)(acc[dt][4 * g + 4] * scale), (bf16)(acc[dt][4 * g + 5] * scale), (bf16)(acc[dt][4 * g + 6] * scale), (bf16)(acc[dt][4 * g + 7] * scale)};
            const uint2 a = __builtin_bit_cast(uint2, a4), bb = __builtin_bit_cast(uint2, b4);
            const auto rx = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
            const unsigned rx0 = rx[0], rx1 = rx[1], ry0 = ry[0], ry1 = ry[1];
            w[gp][0] = rx0; w[gp][1] = ry0; w[gp][2] = rx1; w[gp][3] = ry1;   // channels 16 gp + 8 hi .. + 7 of row (lane & 31)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto r = __builtin_amdgcn_permlane16_swap(w[0][i], w[1][i], false, false);
            const unsigned r0 = r[0], r1 = r[1];
            t[dt][0][i] = r0; t[dt][1][i] = r1;   // rows half*16 + (lane & 15), chunk 2 * ((lane >> 4) & 1) + (lane >> 5) of channel tile dt
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        unsigned z0[4], z1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            z0[i] = dpp_ror8<0xC>(t[0][half][i], t[1][half][i]);   // lanes 0-7 of a row: tile 0 of rows 0-7; lanes 8-15: tile 1 of rows 0-7
            z1[i] = dpp_ror8<0x3>(t[1][half][i], t[0][half][i]);   // lanes 0-7: tile 0 of rows 8-15; lanes 8-15: tile 1 of rows 8-15
        }
        pk[half * 2] = make_uint4(z0[0], z0[1], z0[2], z0[3]);
        pk[half * 2 + 1] = make_uint4(z1[0], z1[1], z1[2], z1[3]);
    }
}
// piece p of a packed tile; rp[p] = its global row, already advanced to this lane's chunk
__device__ __forceinline__ void store_piece(bf16* const (&rp)[4], const bool (&ok)[4], const uint4& v, int p) {
    if (ok[p]) *reinterpret_cast<uint4*>(rp[p]) = v;
}

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
// OUT3 (the complete_model pass of "fp16x3q", whose attention is the hi * hi product of the q / k / v planes): `out` is the proj GEMM's
// operand image in the hi16 / fp8 form, rows of SPLIT_A * 768 16-bit elements [hi | e4m3(hi) | e4m3(lo 2^12)] (store4_split_f8), written
// from the fp32 result: hi = its 16-bit rounding (also what the 16-bit backward reads as o), lo the remainder.  The lo plane goes through
// the same lane exchanges as a second packed tile.
__device__ __forceinline__ uint2 e4m3x8(const uint4& h8) {   // 8 packed 16-bit values -> 8 e4m3 bytes
    const bf16x8 x = __builtin_bit_cast(bf16x8, h8);
    uint2 r;
    r.x = (unsigned)pack4_e4m3((float)x[0], (float)x[1], (float)x[2], (float)x[3]);
    r.y = (unsigned)pack4_e4m3((float)x[4], (float)x[5], (float)x[6], (float)x[7]);
    return r;
}
template <bool PIPE, int ABL = 0, bool OUT3 = false>
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
        __builtin_amdgcn_s_setprio(3);   // the youngest wave of its SIMD would lose every arbitration to the compute wave it shares it with
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
            float mt = kt < 6 ? acc_max16(s) : acc_max8(s);
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
        // O^T tile -> whole 128-byte rows of `out` (pack_rows above)
        const float inv = 1.0f / sum;
        f32x16 on[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { on[0][r] = o[0][r]; on[1][r] = o[1][r]; }
        uint4 pk[4];
        const int chunk = 4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + hi;
        if constexpr (OUT3) {
            f32x16 lo[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = on[dt][r] * inv;
                    asm("" : "+v"(x));   // (split2: one rounding of ONE value)
                    on[dt][r] = x;
                    lo[dt][r] = (x - (float)(bf16)x) * F8_LO_SCALE;
                }
            uint4 pl[4];
            pack_rows(pk, on, 1.0f);
            pack_rows(pl, lo, 1.0f);
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int tr_ = wave * 32 + pc * 8 + (lane & 7);
                if (tr_ >= NT) continue;
                bf16* rowp = out + ((size_t)b * NT + tr_) * (SPLIT_A * D);
                const int col = h * HD + chunk * 8;
                *reinterpret_cast<uint4*>(rowp + col) = pk[pc];
                unsigned char* r8 = reinterpret_cast<unsigned char*>(rowp) + 2 * (size_t)D + col;
                *reinterpret_cast<uint2*>(r8) = e4m3x8(pk[pc]);
                *reinterpret_cast<uint2*>(r8 + D) = e4m3x8(pl[pc]);
            }
            continue;
        }
        pack_rows(pk, on, inv);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int tr_ = wave * 32 + pc * 8 + (lane & 7);
            if (tr_ < NT)
                *reinterpret_cast<uint4*>(out + ((size_t)b * NT + tr_) * D + h * HD + chunk * 8) = pk[pc];
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward: dQ, dK, dV of one (image, head) in one persistent workgroup pass
// ------------------------------------------------------------------------------------------
// The arithmetic of attention.hip's fused kernel (phase B: wave w owns KEY tile w and walks the query tiles -> dK, dV; phase A: wave w
// owns QUERY tile w and walks the key tiles -> dQ; S and dP are recomputed in both), on a different data path:
//   * Q, dO, K, V are four swizzled LDS images filled by LDS-DMA; every transposed operand (dO^T, Q^T, K^T) is a ds_read_b64_tr_b16 of
//     the same image -- no register staging, no transposed copies, no re-staging between the phases;
//   * wave 7 is the loader.  K / V are double-buffered (head i+1's land while head i is computed); Q / dO are single-buffered and
//     refilled as soon as phase B is over (barrier c), i.e. under phase A.  The loader also loads the rows of O and the log-sum-exp of
//     the next head and leaves delta = rowsum(dO o) and lse log2(e) in LDS (pad rows: +inf / 0, which zeroes P and dS of the padding
//     queries without a select) -- the compute waves issue no loads at all, so their dQ / dK / dV stores stay in flight across heads;
//   * images are 208 rows (6 x 26 KB + 3.5 KB of row statistics = 159.5 KB of the 160 KB): tile 6 reads rows 192 .. 223, i.e. 16 rows
//     of the NEXT image (finite data) or past the allocation (zeros); whatever they hold only reaches scores of padding rows / keys.
constexpr int BROWS = 208;
constexpr int BIMG = BROWS * 128;               // 26624
constexpr int BSTAT = 2 * 2 * 224 * 4;          // lse2[2][224], delta[2][224]
constexpr int BWD_LDS = BSTAT + 6 * BIMG;       // 163328

__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ unsigned long long g_bwd_dbg[8 + 16];   // ABL & 8 (probe builds): cycles of wave 0 in barrier (a) / phase B / barrier (c) / phase A / dq stores, heads
// ABL / SETPRIO: measurement variants instantiated by tools/probes/r5/av2_test.hip only (ABL: 1 no stores, 2 no compute, 4 no DMA after the first
// head, 8 per-wave cycle stamps, 16 / 32 / 64 loader ablations; SETPRIO 1 static priority for waves 4-6, 2 alternating per tile, 3 around the
// S / dP MFMA block: all measured equal or worse than none).  The product launches <0, 0, 1>.
template <int ABL = 0, int SETPRIO = 0, int LT = 1>
__global__ __launch_bounds__(512, 2) void attn_bwd_v2_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                             const bf16* __restrict__ v, const bf16* __restrict__ o,
                                                             const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                             bf16* __restrict__ dqkv, int nheads, int nq, int o_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [lse2[2][224] | delta[2][224] | Q | dO | K0 | V0 | K1 | V1]
    float* lse_s = reinterpret_cast<float*>(smem);
    float* del_s = lse_s + 2 * 224;
    char* img0 = smem + BSTAT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int l31 = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 6 * BIMG / 16; i += 512) reinterpret_cast<uint4*>(img0)[i] = make_uint4(0, 0, 0, 0);   // pad rows stay zero for good
    for (int i = tid; i < 2 * 224; i += 512) { lse_s[i] = INFINITY; del_s[i] = 0.f; }
    __syncthreads();
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(img0));
    // Image slots.  A head uses four (Q, dO, K, V) and two are spare.  At barrier (a) of head i the spare pair takes Q / dO of head i+1 (the
    // loader then has the whole of phase B for their row statistics); at barrier (c) the slots of Q_i / dO_i are free and take K / V of
    // head i+1, which have phase A to land; K_i / V_i's slots are the next spare pair.  Loader and compute waves rotate the same indices.
    int sQ = 0, sD = 1, sK = 2, sV = 3, sE = 4, sF = 5;
    auto rotate = [&]() { const int q0 = sQ, d0 = sD, k0 = sK, v0 = sV; sQ = sE; sD = sF; sK = q0; sV = d0; sE = k0; sF = v0; };
    if (wave_s == 7) {   // ---- loader ----
        __builtin_amdgcn_s_setprio(3);   // the youngest wave of its SIMD loses every arbitration to the compute wave it shares it with; its few instructions gate every head
        // rows of O and the log-sum-exp of a head into registers (25 x 16 B + 4 floats per lane), consumed by make_stats
        bf16x8 orow[PIECES];
        float lrow[4];
        const int r8 = lane >> 3, c = lane & 7;
        // LT = 1 (default): asm loads + a COUNTED wait, so that the statistics run under the flight of the K / V pieces issued after them (hipcc
        // would count only its own 29 loads among the ~130 operations in flight and wait for nearly all of them before the first use): 101.5 vs
        // 107.4 us, bit-identical to LT = 0 (full drain, statistics before barrier (c)) over repeated launches at B = 128
        // (tools/probes/r5/av2_test.hip).  The workgroup owns its CU (159.5 KB of LDS), so no other stream's waves share it -- the condition
        // under which DESIGN.md 7b saw a partial wait consumed early.
        auto load_rows = [&](int bh) {
            const int b = bh / NH, h = bh - b * NH;
#pragma unroll
            for (int it = 0; it < PIECES; ++it) {
                const bf16* gp = o + ((size_t)b * NT + min(it * 8 + r8, NT - 1)) * o_ld + h * HD + c * 8;
                if (LT) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(orow[it]) : "v"(gp) : "memory");
                else orow[it] = *reinterpret_cast<const bf16x8*>(gp);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* gp = lse + (size_t)bh * NT + min(lane + 64 * j, NT - 1);
                if (LT) asm volatile("global_load_dword %0, %1, off" : "=v"(lrow[j]) : "v"(gp) : "memory");
                else lrow[j] = *gp;
            }
        };
#define DYT_ROWS_LANDED(N)                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                                  \
                 : "+v"(orow[0]), "+v"(orow[1]), "+v"(orow[2]), "+v"(orow[3]), "+v"(orow[4]), "+v"(orow[5]), "+v"(orow[6]), "+v"(orow[7]),   \
                   "+v"(orow[8]), "+v"(orow[9]), "+v"(orow[10]), "+v"(orow[11]), "+v"(orow[12]), "+v"(orow[13]), "+v"(orow[14]),           \
                   "+v"(orow[15]), "+v"(orow[16]), "+v"(orow[17]), "+v"(orow[18]), "+v"(orow[19]), "+v"(orow[20]), "+v"(orow[21]),         \
                   "+v"(orow[22]), "+v"(orow[23]), "+v"(orow[24]), "+v"(lrow[0]), "+v"(lrow[1]), "+v"(lrow[2]), "+v"(lrow[3])             \
                 :: "memory")
        auto issue_qdo = [&](int bh, int slq, int sld) {
            const int b = bh / NH, h = bh - b * NH;
            dma_image<1>(q + (size_t)bh * NT * HD, HD, lds0 + slq * BIMG, 0, lane);
            dma_image<1>(dout + (size_t)b * NT * D + h * HD, D, lds0 + sld * BIMG, 0, lane);
        };
        auto issue_kv = [&](int bh, int slk, int slv) {
            dma_image<1>(k + (size_t)bh * NT * HD, HD, lds0 + slk * BIMG, 0, lane);
            dma_image<1>(v + (size_t)bh * NT * HD, HD, lds0 + slv * BIMG, 0, lane);
        };
        auto make_stats = [&](int ab, int sld) {   // delta = rowsum(dO o) of the head whose rows load_rows() fetched and whose dO image sits in slot sld
            const char* dOi = img0 + sld * BIMG;
#pragma unroll
            for (int it = 0; it < PIECES; ++it) {
                const int row = it * 8 + r8;
                const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dOi + row * 128 + ((c ^ swz(row)) << 4));
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc = fmaf((float)orow[it][i], (float)dv[i], acc);
                acc += dpp_f32<0xB1>(acc);    // lane ^ 1
                acc += dpp_f32<0x4E>(acc);    // lane ^ 2
                acc += dpp_f32<0x141>(acc);   // row_half_mirror: the other quad of the 8-lane row group
                if (c == 0 && row < NT) del_s[ab * 224 + row] = acc;
                if (it % 5 == 4) __builtin_amdgcn_sched_barrier(0);   // keeps the scheduler from hoisting all 25 image reads (100 registers) to the top
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane + 64 * j < NT) lse_s[ab * 224 + lane + 64 * j] = lrow[j] * LOG2E;
        };
        int bh = blockIdx.x, ab = 0;
        if (bh < nheads) {
            load_rows(bh);
            issue_qdo(bh, sQ, sD);
            issue_kv(bh, sK, sV);
            DYT_ROWS_LANDED(0);
            make_stats(0, sD);
        }
        for (; bh < nheads; bh += gridDim.x, ab ^= 1) {
            const int nb = bh + gridDim.x;
            const bool more = nb < nheads && !(ABL & 4);
            barrier_lds();   // (a) head bh may start; every wave has left head bh - gridDim.x: the spare slots take Q / dO of the next head
            if (more) {
                if (!(ABL & 32)) load_rows(nb);
                issue_qdo(nb, sE, sF);
                if (!LT) {
                    if (!(ABL & 64)) DYT_ROWS_LANDED(0);
                    if (!(ABL & 16)) make_stats(ab ^ 1, sF);
                }
            }
            barrier_lds();   // (c) phase B of head bh is over: the slots of its Q / dO images are free and take K / V of the next head
            if (more) {
                issue_kv(nb, sQ, sD);   // 50 pieces
                if (LT) {
                    DYT_ROWS_LANDED(50);      // everything older than those 50: o rows, lse, dO and Q images (issued a phase ago)
                    make_stats(ab ^ 1, sF);   // ... under the flight of K / V
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            rotate();
        }
#undef DYT_ROWS_LANDED
        return;
    }
    // ---- compute waves ----
    const FragAddr fa(lane);
    const int trow = wave * 32 + l31;   // the wave's key (phase B) / query (phase A) of this lane
    if (SETPRIO == 1 && wave_s >= 4) __builtin_amdgcn_s_setprio(1);   // the second wave of every SIMD wins arbitration: the pair falls into complementary MFMA / VALU phases
    int buf = 0;
    uint4 pq[4];              // packed dQ of the previous head, stored from inside this head's phase B loop
    bf16* pq_row[4] = {nullptr, nullptr, nullptr, nullptr};   // its rows (null: nothing pending)
    bool pq_ok[4] = {false, false, false, false};
    for (int bh = blockIdx.x; bh < nheads; bh += gridDim.x, buf ^= 1, rotate()) {
        const int b = bh / NH, h = bh - b * NH;
        const char* Qi = img0 + sQ * BIMG;
        const char* dOi = img0 + sD * BIMG;
        const char* Ki = img0 + sK * BIMG;
        const char* Vi = img0 + sV * BIMG;
        const float* L2 = lse_s + buf * 224;
        const float* Dl = del_s + buf * 224;
        unsigned long long t0 = 0;
        if (ABL & 8) t0 = __builtin_readcyclecounter();
        barrier_lds();   // (a)
        if (ABL & 8) { const unsigned long long t1 = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_bwd_dbg[0], t1 - t0); t0 = t1; }
        bf16* orow[4];   // rows wave*32 + p*8 + (lane & 7) of dqkv, at this lane's 16-byte chunk of the head's 64 channels
        bool ook[4];
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int tr_ = wave * 32 + pc * 8 + (lane & 7);
            ook[pc] = tr_ < NT;
            orow[pc] = dqkv + ((size_t)b * NT + min(tr_, NT - 1)) * (3 * D) + h * HD + (4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + hi) * 8;
        }
        uint4 pkK[4], pkV[4];   // packed dK / dV of this head, stored from inside the phase A loop
        bf16* const okrow[4] = {orow[0] + D, orow[1] + D, orow[2] + D, orow[3] + D};
        bf16* const ovrow[4] = {orow[0] + 2 * D, orow[1] + 2 * D, orow[2] + 2 * D, orow[3] + 2 * D};
        // ---------------- phase B: dK, dV of key tile `wave` ----------------
        if (!(ABL & 2)) {
            bf16x8 kf[4], vf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { kf[ks] = row_frag(Ki, fa.row[ks] + wave * 4096); vf[ks] = row_frag(Vi, fa.row[ks] + wave * 4096); }
            f32x16 aK[2], aV[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { aK[0][r] = 0.f; aK[1][r] = 0.f; aV[0][r] = 0.f; aV[1][r] = 0.f; }
#pragma unroll 1
            for (int qt = 0; qt < nq; ++qt) {
                if (SETPRIO == 2) { if ((qt ^ (wave_s >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                if (SETPRIO == 3) __builtin_amdgcn_s_setprio(1);
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = MFMA32(row_frag(Qi, fa.row[ks] + qt * 4096), kf[ks], s);      // S[q][key]: q in registers, key = lane
                    dp = MFMA32(row_frag(dOi, fa.row[ks] + qt * 4096), vf[ks], dp);   // dP[q][key]
                }
                if (SETPRIO == 3) __builtin_amdgcn_s_setprio(0);
                // q = qt*32 + 8g + 4hi + e for register 4g + e; the padding queries have lse2 = +inf (P = 0) and delta = 0
                f32x16 p;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 L4 = *reinterpret_cast<const float4*>(L2 + qt * 32 + 8 * g + 4 * hi);
                    const float4 D4 = *reinterpret_cast<const float4*>(Dl + qt * 32 + 8 * g + 4 * hi);
                    const float Ls[4] = {L4.x, L4.y, L4.z, L4.w}, Ds[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -Ls[e]));
                        p[r] = pv;
                        s[r] = pv * (dp[r] - Ds[e]);   // dS
                    }
                }
                if (pq_row[0] && qt < 4) {   // wave-uniform; constant register indices
                    if (qt == 0) store_piece(pq_row, pq_ok, pq[0], 0);
                    else if (qt == 1) store_piece(pq_row, pq_ok, pq[1], 1);
                    else if (qt == 2) store_piece(pq_row, pq_ok, pq[2], 2);
                    else store_piece(pq_row, pq_ok, pq[3], 3);
                }
                const int nh = qt < 6 ? 2 : 1;   // queries 208 .. 223 are all padding
                for (int half = 0; half < nh; ++half) {
                    const bf16x8 pf = half ? pack8(p, 8) : pack8(p, 0);
                    const bf16x8 dsf = half ? pack8(s, 8) : pack8(s, 0);
                    const int off = qt * 4096 + half * 2048;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        aV[dt] = MFMA32(tr_frag(dOi, fa.tr[dt][0] + off, fa.tr[dt][1] + off), pf, aV[dt]);   // dV^T[d][key] += dO^T[d][q] P[q][key]
                        aK[dt] = MFMA32(tr_frag(Qi, fa.tr[dt][0] + off, fa.tr[dt][1] + off), dsf, aK[dt]);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
                    }
                }
            }
            if (pq_row[0]) {   // a short phase B (nq < 4): the rest of the previous head's dQ
                if (nq < 2) store_piece(pq_row, pq_ok, pq[1], 1);
                if (nq < 3) store_piece(pq_row, pq_ok, pq[2], 2);
                if (nq < 4) store_piece(pq_row, pq_ok, pq[3], 3);
            }
            pack_rows(pkK, aK, 1.0f);
            pack_rows(pkV, aV, 1.0f);
            if ((ABL & 1) && aK[0][0] == 123.456f) dqkv[0] = (bf16)(aK[1][3] + aV[0][2] + aV[1][7]);
        }
        // ---------------- phase A: dQ of query tile `wave` ----------------
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = row_frag(Qi, fa.row[ks] + wave * 4096); dof[ks] = row_frag(dOi, fa.row[ks] + wave * 4096); }
        const float L = L2[trow], dl = Dl[trow];
        if (ABL & 8) { const unsigned long long t1 = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_bwd_dbg[1], t1 - t0); if (lane == 0) atomicAdd(&g_bwd_dbg[8 + wave], t1 - t0); t0 = t1; }
        barrier_lds();   // (c): every wave holds its q / dO rows; the loader refills the Q / dO images
        if (ABL & 8) { const unsigned long long t1 = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_bwd_dbg[2], t1 - t0); t0 = t1; }
        f32x16 dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
        if (!(ABL & 2) && wave < nq) {
#pragma unroll
            for (int kt = 0; kt < 7; ++kt) {
                if (SETPRIO == 2) { if ((kt ^ (wave_s >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = MFMA32(row_frag(Ki, fa.row[ks] + kt * 4096), qf[ks], s);      // S^T[key][q]
                    dp = MFMA32(row_frag(Vi, fa.row[ks] + kt * 4096), dof[ks], dp);   // dP^T[key][q]
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -L));
                    s[r] = pv * (dp[r] - dl);   // dS^T
                    if (kt == 6 && 192 + (r & 3) + 8 * (r >> 2) + 4 * hi >= NT) s[r] = 0.f;
                }
                const int nh = kt < 6 ? 2 : 1;   // keys 208 .. 223 are all padding
#pragma unroll
                for (int half = 0; half < nh; ++half) {
                    const bf16x8 dsf = pack8(s, half * 8);
                    const int off = kt * 4096 + half * 2048;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        dq[dt] = MFMA32(tr_frag(Ki, fa.tr[dt][0] + off, fa.tr[dt][1] + off), dsf, dq[dt]);   // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
                }
                if (!(ABL & 3)) {   // 8 pieces of dK / dV over the 7 key tiles
                    if (kt < 4) store_piece(okrow, ook, pkK[kt & 3], kt);
                    else store_piece(ovrow, ook, pkV[kt - 4], kt - 4);
                    if (kt == 6) store_piece(ovrow, ook, pkV[3], 3);
                }
            }
        } else if (!(ABL & 3)) {   // no phase A for this wave (cls-only tail): all eight pieces now
#pragma unroll
            for (int i = 0; i < 4; ++i) { store_piece(okrow, ook, pkK[i], i); store_piece(ovrow, ook, pkV[i], i); }
        }
        if (ABL & 8) { const unsigned long long t1 = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_bwd_dbg[3], t1 - t0); if (lane == 0) atomicAdd(&g_bwd_dbg[16 + wave], t1 - t0); t0 = t1; }
        pack_rows(pq, dq, 0.125f);
        if (!(ABL & 1)) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) { pq_row[pc] = orow[pc]; pq_ok[pc] = ook[pc]; }
        }
        else if (dq[0][0] == 123.456f) dqkv[1] = (bf16)(dq[1][3]);
        if (ABL & 8) { const unsigned long long t1 = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_bwd_dbg[4], t1 - t0); atomicAdd(&g_bwd_dbg[5], 1ull); } }
    }
    if (pq_row[0]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_piece(pq_row, pq_ok, pq[i], i);
    }
}

}  // namespace av2

static int set_lds_v2(const void* fn, size_t bytes) {
    DYT_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

int launch_attn_fwd_v2(const void* q, const void* k, const void* v, void* out, float* lse, int batch, hipStream_t s, int out3_f8) {
    const int grid = batch * NH;
    const size_t lds = 5 * av2::IMG;
    static bool done[64] = {};
    static const int pipe = getenv("DYT_ATTN_V2_PIPE") ? atoi(getenv("DYT_ATTN_V2_PIPE")) : 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (!done[dev & 63]) {
        if (set_lds_v2((const void*)av2::attn_fwd_v2_kernel<true>, lds) || set_lds_v2((const void*)av2::attn_fwd_v2_kernel<false>, lds) ||
            set_lds_v2((const void*)av2::attn_fwd_v2_kernel<true, 0, true>, lds)) return -2;
        done[dev & 63] = true;
    }
    if (out3_f8) {   // `out` = the proj GEMM's operand image in the hi16 / fp8 form
        hipLaunchKernelGGL((av2::attn_fwd_v2_kernel<true, 0, true>), dim3(min(grid, 256)), dim3(512), lds, s, (const bf16*)q, (const bf16*)k, (const bf16*)v,
                           (bf16*)out, lse, grid);
        DYT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    auto* kern = pipe ? av2::attn_fwd_v2_kernel<true> : av2::attn_fwd_v2_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(min(grid, 256)), dim3(512), lds, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, lse, grid);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attn_bwd_v2(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, void* dqkv,
                       int batch, hipStream_t s, int q_tiles, int out_ld) {
    const int grid = batch * NH;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (!done[dev & 63]) {
        if (set_lds_v2((const void*)av2::attn_bwd_v2_kernel<0>, av2::BWD_LDS)) return -2;
        done[dev & 63] = true;
    }
    hipLaunchKernelGGL(av2::attn_bwd_v2_kernel<0>, dim3(min(grid, 256)), dim3(512), av2::BWD_LDS, s, (const bf16*)q, (const bf16*)k, (const bf16*)v,
                       (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, grid, q_tiles, out_ld);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dyt
