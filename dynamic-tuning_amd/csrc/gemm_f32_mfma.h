// C[M,N] = A[M,K] @ W[N,K]^T in EXACT fp32 on the matrix cores (included by gemm.hip): the parity mode's GEMM.
//
// v_mfma_f32_32x32x2_f32 takes fp32 operands and accumulates in fp32 with one rounding per product -- bit for bit a
// k-ordered fmaf chain (MI355X_MICROARCH.md, "Matrix cores") -- at the fp32 vector peak (64 FLOP/clk/SIMD, 157 TFLOP/s),
// but from ONE instruction per 64 cycles, which leaves the VALU, the LDS and the memory pipes almost idle: a vector-ALU
// fp32 GEMM needs every issue slot for its v_fma and stalls on its own operand traffic (the round-1 64x64x16 kernel ran
// at a third of the peak).
//
// Tile 128 x BN x 32 (BN = 128 or 64), 4 waves (2x2), wave tile 64 x BN/2 = 2 x TN MFMA tiles of 32x32.
//   * LDS image = the bf16 kernels' geometry: one operand row of a K stage is 32 fp32 = 128 B = 8 chunks of 16 B (a
//     wave-instruction moves 8 rows = 1 KiB), staged HBM -> registers -> LDS into a 2-slot ring.
//   * The operand of a 32x32x2 MFMA is ONE fp32 per lane: lane l holds row (l & 31), k = l >> 5.  A lane fetches a whole
//     16-B chunk (4 consecutive k) with one ds_read_b128 -- lanes 0..31 chunk 2j, lanes 32..63 chunk 2j+1 -- and feeds
//     register t of it to MFMA t: that MFMA multiplies k = 8j + t (low half) and k = 8j + 4 + t (high half).  A and W
//     fragments use the same assignment, so the products pair up; only the ORDER in which the 32 k of a stage are
//     accumulated is permuted (still one fp32 fma per product).
//   * chunk slot XOR-swizzled with ((row >> 1) & 7) (on the global source address and on the ds_read): the 32 rows x 16 B
//     a half-wave reads in one ds_read_b128 then cover every bank group exactly twice per 16-lane service group = the
//     conflict-free pattern for that instruction (rows 2i, 2i+1 share a swizzle value and differ in the 128-B half of the
//     256-B bank line).
//   * W goes in as the MFMA's A operand, so a lane ends up with 4 CONSECUTIVE output columns of one row per register
//     quad (D[n][m]: m = lane & 31, n = 8g + 4(lane >> 5) + e for register 4g + e), which is what the LDS-staged
//     epilogue (same scheme and functors as the bf16 kernels) parks as one ds_write_b128.
// Per K stage and wave: 64 MFMAs (4096 matrix-pipe cycles), 16 ds_read_b128, 8 global_load_dwordx4 + 8 ds_write_b128
// (BN = 128), all slotted between the MFMAs (see the main loop); everything but the matrix pipe has > 10x slack.
#pragma once

namespace dyt {

template <int BM, int BN, int WAVES_M, int WAVES_N, class Epi>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void gemm_f32_mfma_nt_kernel(
    const float* __restrict__ A, const float* __restrict__ W, int M, int N, int K, const int* __restrict__ m_dev,
    const int* __restrict__ a_map, int m_begin, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = 32;                        // fp32 per row and stage = 128 B
    constexpr int NW = WAVES_M * WAVES_N, NTHR = 64 * NW;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && WM % 32 == 0 && WN % 32 == 0, "tile/wave layout");

    const int Mv = m_dev ? min(*m_dev, M) : M;
    // XCD-aware bijective block remap (see gemm_bf16_nt_kernel): consecutive logical tiles share an A row panel
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tiles_n = N / BN;
    const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
    const int m0 = m_begin + tm * BM, n0 = tn * BN;
    if (m0 >= Mv) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;

    // ---- staging: lane -> (row-in-8, 16-B slot); source chunk = slot ^ ((tile row >> 1) & 7) ----
    const int lrow = lane >> 3, slot = lane & 7;
    const float* a_src[A_INSTR];
    const float* b_src[B_INSTR];
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) {
        const int row = (t * NW + wave) * 8 + lrow;
        int grow = min(m0 + row, Mv - 1);
        if (a_map) grow = a_map[grow];   // gathered A rows (compacted MLP backward)
        a_src[t] = A + (size_t)grow * K + ((slot ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int t = 0; t < B_INSTR; ++t) {
        const int row = (t * NW + wave) * 8 + lrow;
        b_src[t] = W + (size_t)(n0 + row) * K + ((slot ^ ((row >> 1) & 7)) << 2);
    }
    // one K stage of this wave's share of the tile, HBM -> registers -> LDS.  (Not LDS-DMA as in the bf16 kernels: a
    // global_load_lds piece costs 60-185 issue cycles in the wave's instruction stream -- more than the 64-cycle shadow of
    // one fp32 MFMA, measured as 18 % idle matrix pipe -- where a global_load_dwordx4 + ds_write_b128 pair costs ~20, and
    // at 1/16 of the bf16 operand rate the extra LDS write bandwidth is irrelevant.)
    f32x4 g[A_INSTR + B_INSTR];
    auto gload = [&](int kt) {
#pragma unroll
        for (int t = 0; t < A_INSTR; ++t) g[t] = *reinterpret_cast<const f32x4*>(a_src[t] + kt * BK);
#pragma unroll
        for (int t = 0; t < B_INSTR; ++t) g[A_INSTR + t] = *reinterpret_cast<const f32x4*>(b_src[t] + kt * BK);
    };
    auto gstore = [&](int buf) {
        char* base = smem + buf * STAGE + lane * 16;
#pragma unroll
        for (int t = 0; t < A_INSTR; ++t) *reinterpret_cast<f32x4*>(base + (t * NW + wave) * 1024) = g[t];
#pragma unroll
        for (int t = 0; t < B_INSTR; ++t) *reinterpret_cast<f32x4*>(base + A_BYTES + (t * NW + wave) * 1024) = g[A_INSTR + t];
    };

    // ---- fragment read offsets: row = tile base + (lane & 31); round j reads chunk 2j + (lane >> 5) ----
    const int l31 = lane & 31, hi = lane >> 5;
    const int fsw = (l31 >> 1) & 7;   // tile bases are multiples of 32: the swizzle value depends on the lane only
    const int a_off = (wm * WM + l31) * 128;
    const int b_off = A_BYTES + (wn * WN + l31) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;   // even (K % 64 == 0 is checked at launch)
    // Software pipeline over a 2-slot LDS ring, two fragment register sets and one staging register set.  Stage kt:
    //   first quarter of its 64 MFMAs : park the staged registers (stage kt+1, loaded a stage ago) in slot (kt+1)&1 and
    //                                   request stage kt+2 from HBM (a whole stage to land)
    //   barrier after the third quarter: the writes of stage kt+1 are visible
    //   last quarter                   : read every fragment of stage kt+1 into the other register set
    // so the matrix pipe only ever waits for the barrier skew.  Slot (kt+1)&1 last held stage kt-1, whose fragments every
    // wave read (and consumed) before it arrived at the barrier of stage kt-1 -- which the writer has passed.  Past the
    // end the last stage is re-staged and the stale slot re-read (unused): keeps the loop body branch-free.
    f32x4 afA[4][TM], wfA[4][TN], afB[4][TM], wfB[4][TN];
    constexpr int NMQ = 4 * TM * TN;                                    // MFMAs per quarter
    constexpr int NG = A_INSTR + B_INSTR;
    constexpr int RPM = (4 * (TM + TN) + NMQ - 1) / NMQ;                // fragment reads behind each MFMA of the last quarter
    constexpr int GPM = (NG + NMQ / 2 - 1) / (NMQ / 2);                 // LDS writes / global loads behind each MFMA of a half of the first quarter
#define DYT_F_READ(AF, WF, BUF)                                                                          \
    {                                                                                                    \
        const char* rb_ = smem + (BUF) * STAGE;                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
            const int so = ((2 * j + hi) ^ fsw) << 4;                                                    \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                               \
                AF[j][i] = *reinterpret_cast<const f32x4*>(rb_ + a_off + i * 4096 + so);                 \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                               \
                WF[j][i] = *reinterpret_cast<const f32x4*>(rb_ + b_off + i * 4096 + so);                 \
        }                                                                                                \
    }
#define DYT_F_MMA(AF, WF, J0, J1)                                                                        \
    _Pragma("unroll") for (int j = J0; j < J1; ++j) _Pragma("unroll") for (int t = 0; t < 4; ++t)        \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jn = 0; jn < TN; ++jn) \
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(WF[j][jn][t], AF[j][i][t], acc[i][jn], 0, 0, 0);
// one stage: CUR = ring slot of stage kt (fragments already in AFC / WFC)
#define DYT_F_STAGE(AFC, WFC, AFN, WFN, CUR)                                                             \
    {                                                                                                    \
        gstore(1 - (CUR));                                                                               \
        gload(min(kt + 2, nk - 1));                                                                      \
        DYT_F_MMA(AFC, WFC, 0, 1)                                                                        \
        _Pragma("unroll") for (int x = 0; x < NMQ / 2; ++x) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x200, GPM, 0);                                         \
        }                                                                                                \
        _Pragma("unroll") for (int x = 0; x < NMQ / 2; ++x) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x020, GPM, 0);                                         \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        DYT_F_MMA(AFC, WFC, 1, 3)                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        __syncthreads();                                                                                 \
        DYT_F_READ(AFN, WFN, 1 - (CUR))                                                                  \
        DYT_F_MMA(AFC, WFC, 3, 4)                                                                        \
        _Pragma("unroll") for (int x = 0; x < NMQ; ++x) {                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);                                         \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        ++kt;                                                                                            \
    }
    gload(0);
    gstore(0);
    gload(min(1, nk - 1));
    __syncthreads();
    DYT_F_READ(afA, wfA, 0)
    for (int kt = 0; kt < nk;) {
        DYT_F_STAGE(afA, wfA, afB, wfB, 0)
        DYT_F_STAGE(afB, wfB, afA, wfA, 1)
    }
#undef DYT_F_STAGE
#undef DYT_F_MMA
#undef DYT_F_READ

    {
        // ---- epilogue through LDS (scheme of gemm_bf16_nt_kernel): the fp32 tile is parked in the staging ring with
        // the 16-B chunk index XOR-swizzled by (row & 7) and read back row-major, so that every global access of the
        // functor is a full-row coalesced transaction.  acc[i][j][4g + e] = C[wm*WM + i*32 + (lane & 31)]
        // [wn*WN + j*32 + 8g + 4(lane >> 5) + e].
        static_assert(BM * BN * 4 <= 2 * STAGE, "epilogue staging does not fit the LDS ring");
        float* Cs = reinterpret_cast<float*>(smem);
        constexpr int CH = BN / 4;
        static_assert(NTHR % CH == 0, "a lane must keep one column chunk for the whole tile");
        constexpr int RSTEP = NTHR / CH, ITERS = BM / RSTEP, BATCH = ITERS % 8 == 0 ? 8 : ITERS;
        const int ch = tid % CH, rl0 = tid / CH;
        const int col = n0 + ch * 4;
        const typename Epi::Col cc = epi.col_init(col);
        constexpr bool PRE_ALL = sizeof(typename Epi::Pre) * ITERS <= 256;
        typename Epi::Pre pr[PRE_ALL ? ITERS : BATCH];
        if (PRE_ALL) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) pr[it] = epi.pre(min(m0 + rl0 + it * RSTEP, Mv - 1), col);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rl = wm * WM + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int chw = ((wn * WN + j * 32) >> 2) + 2 * g + hi;
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(Cs + rl * BN + ((chw ^ (rl & 7)) << 2)) = v;
                }
        }
        __syncthreads();
#pragma unroll
        for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
            if (!PRE_ALL) {
#pragma unroll
                for (int u = 0; u < BATCH; ++u) pr[u] = epi.pre(min(m0 + rl0 + (it0 + u) * RSTEP, Mv - 1), col);
            }
            f32x4 c4[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int rl = rl0 + (it0 + u) * RSTEP;
                c4[u] = *reinterpret_cast<const f32x4*>(Cs + rl * BN + ((ch ^ (rl & 7)) << 2));
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int row = m0 + rl0 + (it0 + u) * RSTEP;
                if (row < Mv) {
                    const float v[4] = {c4[u][0], c4[u][1], c4[u][2], c4[u][3]};
                    epi.apply(row, col, v, cc, pr[PRE_ALL ? it0 + u : u]);
                }
            }
        }
    }
}

}  // namespace dyt
