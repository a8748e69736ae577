// C[M,N] = A[M,K] @ W[N,K]^T with the FROZEN weight pre-shuffled into MFMA fragment order (included by gemm.hip).
//
// Every big GEMM of the DyT path multiplies activations by a frozen weight (forward: W, dgrad: W^T, both known
// when the checkpoint is loaded), so the weight operand can be laid out ONCE exactly as the matrix instruction wants
// it:  Wp[n / 16][k / 32][lane][8]  with lane = ((k % 32) / 8) << 4 | (n % 16)  -- one 16x32 block is 1 KiB in lane
// order.  A wave then fetches a fragment with ONE fully coalesced global_load_dwordx4 straight into registers:
//   * the weight never touches LDS: LDS holds the activation tile only (16 KB per stage instead of 48 KB), the LDS read
//     traffic per MFMA drops by a third (the 8-wave 256x256 kernel runs LDS-bound: 256 KB of LDS traffic per 2048 MFMA
//     cycles) and a 128x256 tile needs 64 KB, so TWO workgroups share a CU: one's epilogue (VALU, stores) overlaps the
//     other's main loop, which a single 128 KB workgroup per CU cannot do;
//   * the wave tile stays 128x64 (4 waves side by side, all sharing the 128 activation rows).
// Per K stage (64) and wave: 64 MFMAs, 16 ds_read_b128 (A), 8 global_load_dwordx4 (W, one ks-block ahead, two
// register sets), 4 LDS-DMA pieces (A, three stages ahead in a 4-slot ring).
#pragma once

namespace dyt {

// W [N,K] row-major bf16 -> fragment order (N % 16 == 0, K % 32 == 0); one thread per 16-byte chunk
__global__ void preshuffle_w_kernel(const bf16* __restrict__ W, bf16* __restrict__ Wp, int N, int K) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kc8 = K / 8;
    if (t >= (int64_t)N * kc8) return;
    const int n = (int)(t / kc8), c = (int)(t - (int64_t)n * kc8);   // c = k / 8
    const int lane = ((c & 3) << 4) | (n & 15);
    const size_t blk = (size_t)(n >> 4) * (K / 32) + (c >> 2);
    *reinterpret_cast<bf16x8*>(Wp + (blk * 64 + lane) * 8) = *reinterpret_cast<const bf16x8*>(W + (size_t)n * K + c * 8);
}

template <class Epi, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_bpre_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Wp, int M,
                                                                int N, int K, const int* __restrict__ m_dev,
                                                                const int* __restrict__ a_map, int m_begin, Epi epi, float out_scale) {
    constexpr int BM = 128, BN = 256, BK = 64, NW = 4, NTHR = 256, SLOTS = 4;
    constexpr int SLOT = BM * BK * 2;            // 16 KB: the activation tile of one K stage
    // One LDS OBJECT per ring slot, and the K loop unrolled over the ring: the compiler's wait-count pass makes every
    // ds_read wait for all LDS-DMA still in flight into an object it may alias (with one array and a computed slot that
    // is always "may alias": a vmcnt(0) at the top of every stage, i.e. the full latency of the pieces issued a
    // moment ago).  Distinct objects are provably disjoint, so fragment reads of slot s proceed while slot s+3 is
    // being filled.
    __shared__ __attribute__((aligned(16))) char ring0[SLOT];
    __shared__ __attribute__((aligned(16))) char ring1[SLOT];
    __shared__ __attribute__((aligned(16))) char ring2[SLOT];
    __shared__ __attribute__((aligned(16))) char ring3[SLOT];
    constexpr bool RS = RowStats<Epi>::value;   // LayerNorm folded in: per-row (mean, rstd) merged from the producer's partials, parked in LDS
    __shared__ float2 ln_s[RS ? BM : 1];
    constexpr int TM = 8, TN = 4, WN = 64;
    constexpr int A_INSTR = BM / 8 / NW;         // 4 LDS-DMA pieces (8 rows x 128 B) per wave and stage

    const int Mv = m_dev ? min(*m_dev, M) : M;
    const int tiles_n = N / BN;
    // XCD-aware remap over the tiles that EXIST: with a device-side row count (compacted student pass: 17 690 of 25 216 rows) the grid is
    // sized for M, and a remap over gridDim hands the XCDs contiguous tile ranges of which the last ones are entirely past Mv -- two of the
    // eight XCDs then idle while the others run as many rounds as the dense launch (round 4: student fc1 154 us vs dense 168).  Blocks
    // are dispatched round-robin over the XCDs, so the first `nwg` block ids cover all eight evenly.
    const int nwg = min((int)gridDim.x, ((max(Mv - m_begin, 0) + BM - 1) / BM) * tiles_n), bid = blockIdx.x;
    if (bid >= nwg) return;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
    const int m0 = m_begin + tm * BM, n0 = tn * BN;
    if (m0 >= Mv) return;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    unsigned long long t_start = 0, t_loop0 = 0, t_loop1 = 0;
    if (ABL == 9) t_start = __builtin_readcyclecounter();

    // ---- A staging (same image as the LDS-staged kernels: [row][8 x 16 B], chunk slot XOR (row & 7)) ----
    const int lrow = lane >> 3, slot8 = lane & 7, chunk = slot8 ^ lrow;
    unsigned a_src[A_INSTR];   // element offsets (M x K < 2^31): half the registers of four 64-bit pointers
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) {
        int grow = min(m0 + (t * NW + wn) * 8 + lrow, Mv - 1);
        if (a_map) grow = a_map[grow];
        a_src[t] = (unsigned)grow * (unsigned)K + chunk * 8;
    }
    auto stage_one = [&](char* ring, int kt, int t) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)a_src[t] + kt * BK),
                                         (__attribute__((address_space(3))) void*)(ring + (t * NW + wn) * 1024), 16, 0, 0);
    };
    // ---- W fragments straight from global: block (n-tile, k-chunk of 32) = 512 elements in lane order ----
    // (the block address is wave-uniform: scalar base + one per-lane offset register)
    const int kblocks = K / 32;
    const int wn_u = __builtin_amdgcn_readfirstlane(wn);
    const bf16* wp[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wp[j] = Wp + ((size_t)((n0 + wn_u * WN) / 16 + j) * kblocks) * 512;
    const int wl = lane * 8, wlb = lane * 16;

    const int frow = lane & 15;
    const int fslot0 = (lane >> 4) ^ (lane & 7), fslot1 = (4 + (lane >> 4)) ^ (lane & 7);
    const int a_off = frow * 128;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK, nb = 2 * nk;
    if (ABL == 9) t_loop0 = __builtin_readcyclecounter();

    bf16x8 faA[TM], faB[TM], fw0[TN], fw1[TN];
    // prologue: stages 0..2 in flight, W block 0, A fragments of (0, ks = 0)
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) stage_one(ring0, 0, t);
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) stage_one(ring1, min(1, nk - 1), t);
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) stage_one(ring2, min(2, nk - 1), t);
#pragma unroll
    for (int j = 0; j < TN; ++j) fw0[j] = *reinterpret_cast<const bf16x8*>(wp[j] + wl);
    if constexpr (RS) {   // behind the first DMA stages (its loads return after them anyway); the barrier below publishes ln_s
        if (tid < BM) {
            int grow = min(m0 + tid, Mv - 1);
            if (a_map) grow = a_map[grow];
            const float2 st = ln_merge_parts(epi.ln_part, grow);
            ln_s[tid] = st;
            if (tn == 0 && m0 + tid < Mv) epi.st_out[grow] = st;
        }
    }
    dma_wait_all();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) faA[i] = *reinterpret_cast<const bf16x8*>(ring0 + a_off + i * 2048 + fslot0 * 16);

#define DYT_P_MMA(FW, FA)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)            \
        acc[i][j] = DYT_MFMA_16x16x32(FW[j], FA[i], acc[i][j]);
#define DYT_P_M2R __builtin_amdgcn_sched_group_barrier(0x008, 2, 2); __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
#define DYT_P_M2V __builtin_amdgcn_sched_group_barrier(0x008, 2, 2); __builtin_amdgcn_sched_group_barrier(0x010, 1, 2);
#define DYT_P_M2 __builtin_amdgcn_sched_group_barrier(0x008, 2, 2);
// one K stage with the ring position known at compile time: CUR holds stage kt, NEXT stage kt+1, FILL receives kt+3
#define DYT_P_STAGE(CUR, NEXT, FILL)                                                                                   \
    {                                                                                                                   \
        /* block A: MFMAs of (kt, ks = 0) on (fw0, faA); fetch W block 2kt+1 -> fw1, A fragments (kt, ks = 1) -> faB */ \
        {                                                                                                               \
            const size_t wo = (size_t)min(2 * kt + 1, nb - 1) * 512;                                                    \
            /* fw0 was requested in the previous block B BEFORE its 4 DMA pieces: at most 4 younger operations may      \
               still be in flight (VMEM returns in order) */                                                           \
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(fw0[0]), "+v"(fw0[1]), "+v"(fw0[2]), "+v"(fw0[3]));               \
            DYT_P_RB(CUR, 0) DYT_P_RB(CUR, 1) DYT_P_W(fw1, 0) DYT_P_RB(CUR, 2) DYT_P_RB(CUR, 3) DYT_P_W(fw1, 1)          \
            DYT_P_RB(CUR, 4) DYT_P_RB(CUR, 5) DYT_P_W(fw1, 2) DYT_P_RB(CUR, 6) DYT_P_RB(CUR, 7) DYT_P_W(fw1, 3)          \
            DYT_P_MMA(fw0, faA)                                                                                         \
            DYT_P_M2R DYT_P_M2R DYT_P_M2V DYT_P_M2 DYT_P_M2R DYT_P_M2R DYT_P_M2V DYT_P_M2                                \
            DYT_P_M2R DYT_P_M2R DYT_P_M2V DYT_P_M2 DYT_P_M2R DYT_P_M2R DYT_P_M2V DYT_P_M2                                \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
        /* block B: stage kt+1 must be visible; MFMAs of (kt, ks = 1) on (fw1, faB); fetch W block 2kt+2 -> fw0, 4 DMA  \
           pieces of stage kt+3 -> FILL, A fragments (kt+1, ks = 0) -> faA */                                          \
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(fw1[0]), "+v"(fw1[1]), "+v"(fw1[2]), "+v"(fw1[3]) : : "memory");      \
        __syncthreads();                                                                                                \
        {                                                                                                               \
            const size_t wo = (size_t)min(2 * kt + 2, nb - 1) * 512;                                                    \
            const int nxt = min(kt + 3, nk - 1);                                                                        \
            /* W loads first, DMA pieces after them (the counted wait at the top of the next stage relies on it) */     \
            DYT_P_W(fw0, 0) DYT_P_RA(NEXT, 0) DYT_P_W(fw0, 1) DYT_P_RA(NEXT, 1) DYT_P_W(fw0, 2) DYT_P_RA(NEXT, 2)        \
            DYT_P_W(fw0, 3) DYT_P_RA(NEXT, 3)                                                                           \
            stage_one(FILL, nxt, 0); DYT_P_RA(NEXT, 4) stage_one(FILL, nxt, 1); DYT_P_RA(NEXT, 5)                        \
            stage_one(FILL, nxt, 2); DYT_P_RA(NEXT, 6) stage_one(FILL, nxt, 3); DYT_P_RA(NEXT, 7)                        \
            DYT_P_MMA(fw1, faB)                                                                                         \
            DYT_P_M2V DYT_P_M2R DYT_P_M2V DYT_P_M2R DYT_P_M2V DYT_P_M2R DYT_P_M2V DYT_P_M2R                              \
            DYT_P_M2V DYT_P_M2R DYT_P_M2V DYT_P_M2R DYT_P_M2V DYT_P_M2R DYT_P_M2V DYT_P_M2R                              \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
        ++kt;                                                                                                           \
    }
// W loads are inline asm: the compiler's own wait insertion counts register loads and LDS-DMA pieces as unordered and
// falls back to vmcnt(0) before the first use, which waits for every DMA piece issued after the load as well
#define DYT_P_W(SET, j) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(SET[j]) : "v"(wlb), "s"(wp[j] + wo));
#define DYT_P_RB(R, i) faB[i] = *reinterpret_cast<const bf16x8*>(R + a_off + (i) * 2048 + fslot1 * 16);
#define DYT_P_RA(R, i) faA[i] = *reinterpret_cast<const bf16x8*>(R + a_off + (i) * 2048 + fslot0 * 16);
    // K % 256 == 0 (768, 2304, 3072 on this path): four stages per trip and NO exit between them -- the four stages must
    // be one basic block, or the compiler sinks the next stage's W loads across the exit edge to just before their use
    for (int kt = 0; kt < nk;) {
        DYT_P_STAGE(ring0, ring1, ring3)
        DYT_P_STAGE(ring1, ring2, ring0)
        DYT_P_STAGE(ring2, ring3, ring1)
        DYT_P_STAGE(ring3, ring0, ring2)
    }
    // the last block B still requested a (clamped) W block: drain it before its destination registers are re-used
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(fw0[0]), "+v"(fw0[1]), "+v"(fw0[2]), "+v"(fw0[3]) : : "memory");
#undef DYT_P_STAGE
#undef DYT_P_W
#undef DYT_P_RB
#undef DYT_P_RA
#undef DYT_P_MMA
#undef DYT_P_M2R
#undef DYT_P_M2V
#undef DYT_P_M2
    if (ABL == 9) t_loop1 = __builtin_readcyclecounter();

    {
        // ---- epilogue through LDS (see gemm_bf16_nt_kernel): the 128x256 fp32 tile goes through the 64 KB ring in two
        // passes of 64 rows; every wave owns all 128 rows of its 64 columns, so all four waves write in both passes.
        constexpr int PASSES = 2, PROWS = BM / PASSES;
        // 64 staged rows x 1 KB: 16 rows per slot array
        auto ringp = [&](int r) -> char* { return r == 0 ? ring0 : (r == 1 ? ring1 : (r == 2 ? ring2 : ring3)); };
        constexpr int CH = BN / 4, RSTEP = NTHR / CH, ITERS = PROWS / RSTEP, BATCH = 8;
        const int ch = tid % CH, rl0 = tid / CH;
        const int col = n0 + ch * 4;
        const typename Epi::Col cc = epi.col_init(col);
        dma_wait_all();
#pragma unroll 1
        for (int p = 0; p < PASSES; ++p) {
            constexpr bool PRE_ALL = sizeof(typename Epi::Pre) * ITERS <= 256;
            typename Epi::Pre pr[PRE_ALL ? ITERS : BATCH];
            if (PRE_ALL) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    if constexpr (RS) pr[it] = ln_s[p * PROWS + rl0 + it * RSTEP];
                    else pr[it] = epi.pre(min(m0 + p * PROWS + rl0 + it * RSTEP, Mv - 1), col);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if ((i >> 2) == p) {
                    const int rl = (i & 3) * 16 + frow;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int chw = ((wn * WN + j * 16) >> 2) + (lane >> 4);
                        *reinterpret_cast<f32x4*>(ringp(i & 3) + frow * 1024 + ((chw ^ (rl & 7)) << 4)) = acc[i][j];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
                if (!PRE_ALL) {
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        if constexpr (RS) pr[u] = ln_s[p * PROWS + rl0 + (it0 + u) * RSTEP];
                        else pr[u] = epi.pre(min(m0 + p * PROWS + rl0 + (it0 + u) * RSTEP, Mv - 1), col);
                    }
                }
                f32x4 c4[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int rl = rl0 + (it0 + u) * RSTEP;
                    c4[u] = *reinterpret_cast<const f32x4*>(ringp(rl >> 4) + (rl & 15) * 1024 + ((ch ^ (rl & 7)) << 4));
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int row = m0 + p * PROWS + rl0 + (it0 + u) * RSTEP;
                    if (row < Mv) {
                        const float v[4] = {c4[u][0] * out_scale, c4[u][1] * out_scale, c4[u][2] * out_scale, c4[u][3] * out_scale};   // 1.0 except in the split fp32 form of a gradient GEMM
                        epi.apply(row, col, v, cc, pr[PRE_ALL ? it0 + u : u]);
                    }
                }
            }
        }
    }
    if (ABL == 9 && tid == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        atomicAdd(&g_gemm_dbg[0], t_loop0 - t_start);
        atomicAdd(&g_gemm_dbg[1], t_loop1 - t_loop0);
        atomicAdd(&g_gemm_dbg[2], t_end - t_loop1);
        atomicAdd(&g_gemm_dbg[3], 1ull);
    }
}

}  // namespace dyt
