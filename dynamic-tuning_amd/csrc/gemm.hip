// Dense contractions of the DyT hot path for gfx950.
//
//   C[M,N] = A[M,K] @ W[N,K]^T  (+ fused epilogue)
//
// fast path  : bf16 operands, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//              128x128x64 (or 128x64x64) tile, 4 waves (2x2), operands staged HBM->LDS by
//              LDS-DMA (global_load_lds_dwordx4) into a 2-deep ring, one barrier per K step.
//              LDS image per operand: [row][8 x 16B chunks], chunk slot XOR-swizzled with
//              (row & 7) on the SOURCE address (the DMA destination is lane-linear), and the
//              same XOR on the ds_read_b128 -- conflict-free for the 16x16x32 fragment read.
//              The MFMA is issued with the W fragment as the A operand so that every lane ends
//              up with 4 CONSECUTIVE output columns of one row (8/16-byte epilogue stores).
// exact path : fp32 operands on the matrix cores (v_mfma_f32_32x32x2_f32, gemm_f32_mfma.h) -- the parity mode.
//
// Reference ops replaced: nn.Linear of Attention.qkv / .proj (models/vision_transformer_IN21K.py:56,73),
// timm Mlp fc1/fc2 (:124-129,159), Adapter.down_proj/up_proj (models/dynamic_adapter.py:124-128),
// PatchEmbed's Conv2d (:272-278) and their autograd dgrads.
#include <type_traits>
#include "kernels.h"

namespace dyt {

// ------------------------------------------------------------------------------------------
// epilogues.  One functor call handles 4 consecutive columns of one output row, in three parts so that
// the MFMA kernel can keep memory latency off the critical path:
//   col_init(col)            per-lane column constants (bias ...): a lane's column is the same for every
//                            chunk it handles, so this is loaded ONCE per tile
//   pre(row, col)            the functor's own global loads (residual, gelu', row maps), returned RAW so
//                            that a whole pass worth of them is in flight before the first use
//   apply(row, col, v, c, p) the arithmetic and the stores
// ------------------------------------------------------------------------------------------
struct NoCtx {};
struct Bias4 { float b[4]; };
__device__ __forceinline__ Bias4 load_bias4(const float* bias, int col) {
    Bias4 r;
    if (bias) { r.b[0] = bias[col]; r.b[1] = bias[col + 1]; r.b[2] = bias[col + 2]; r.b[3] = bias[col + 3]; }
    else { r.b[0] = r.b[1] = r.b[2] = r.b[3] = 0.f; }
    return r;
}
template <class T> struct Raw4;   // 4 consecutive activations as loaded (conversion deferred to the use)
template <> struct Raw4<float> {
    float4 v;
    __device__ __forceinline__ void get(float (&o)[4]) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
template <> struct Raw4<bf16> {
    bf16x4 v;
    __device__ __forceinline__ void get(float (&o)[4]) const {
        o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
    }
};
__device__ __forceinline__ Raw4<float> load_raw4(const float* p) { return {*reinterpret_cast<const float4*>(p)}; }
__device__ __forceinline__ Raw4<bf16> load_raw4(const bf16* p) { return {*reinterpret_cast<const bf16x4*>(p)}; }
// streaming store: the value is not read again soon (gelu' is consumed by the backward pass) -- keeps the
// write burst out of the L2 lines the main loops of the other workgroups are hitting
__device__ __forceinline__ void store4_nt(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    DYT_NT_STORE(v, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ void store4_nt(bf16* p, float a, float b, float c, float d) {
    bf16x4 v = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
    DYT_NT_STORE(v, reinterpret_cast<bf16x4*>(p));
}
struct EpiBiasF32 {
    const float* bias; float* out; int ld;
    typedef Bias4 Col; typedef NoCtx Pre;
    __device__ __forceinline__ Col col_init(int col) const { return load_bias4(bias, col); }
    __device__ __forceinline__ Pre pre(int, int) const { return {}; }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre&) const {
        store4(out + (size_t)row * ld + col, a[0] + c.b[0], a[1] + c.b[1], a[2] + c.b[2], a[3] + c.b[3]);
    }
};

template <class AT>
struct EpiQKV {
    const float* bias; AT* q; AT* k; AT* v;
    struct Col { Bias4 b; AT* base; float s; };
    typedef NoCtx Pre;
    __device__ __forceinline__ Col col_init(int col) const {
        const int which = col / D;
        const int c = col - which * D;
        const int h = c >> 6, d = c & 63;
        Col r;
        r.b = load_bias4(bias, col);
        r.s = which == 0 ? 0.125f : 1.0f;  // head_dim ** -0.5, exact in every dtype
        r.base = (which == 0 ? q : (which == 1 ? k : v)) + (((size_t)h * NT) << 6) + d;
        return r;
    }
    __device__ __forceinline__ Pre pre(int, int) const { return {}; }
    __device__ __forceinline__ void apply(int row, int, const float (&a)[4], const Col& c, const Pre&) const {
        const int b = row / NT, n = row - b * NT;
        AT* dst = c.base + (((size_t)b * NH * NT + n) << 6);
        store4(dst, (a[0] + c.b.b[0]) * c.s, (a[1] + c.b.b[1]) * c.s, (a[2] + c.b.b[2]) * c.s, (a[3] + c.b.b[3]) * c.s);
    }
};

// q / k / v of the split forms as two 16-bit planes each (hi = rn16(x), lo = rn16(x - hi), layout [B*12][197][64] like the fp32 tensors they
// replace, the same bytes): the split attention kernel stages them without conversion work and the hi planes ARE the 16-bit q / k / v a
// 16-bit backward pass reads (dyt_ctx::bwd16)
struct EpiQKVPlanes {
    const float* bias; bf16 *qh, *kh, *vh, *ql, *kl, *vl;
    struct Col { Bias4 b; bf16* bh; bf16* bl; float s; };
    typedef NoCtx Pre;
    __device__ __forceinline__ Col col_init(int col) const {
        const int which = col / D;
        const int c = col - which * D;
        const int h = c >> 6, d = c & 63;
        Col r;
        r.b = load_bias4(bias, col);
        r.s = which == 0 ? 0.125f : 1.0f;
        const size_t off = (((size_t)h * NT) << 6) + d;
        r.bh = (which == 0 ? qh : (which == 1 ? kh : vh)) + off;
        r.bl = (which == 0 ? ql : (which == 1 ? kl : vl)) + off;
        return r;
    }
    __device__ __forceinline__ Pre pre(int, int) const { return {}; }
    __device__ __forceinline__ void apply(int row, int, const float (&a)[4], const Col& c, const Pre&) const {
        const int b = row / NT, n = row - b * NT;
        const size_t o = ((size_t)b * NH * NT + n) << 6;
        const Split2 s0 = split2((a[0] + c.b.b[0]) * c.s), s1 = split2((a[1] + c.b.b[1]) * c.s), s2 = split2((a[2] + c.b.b[2]) * c.s),
                     s3 = split2((a[3] + c.b.b[3]) * c.s);
        *reinterpret_cast<bf16x4*>(c.bh + o) = bf16x4{s0.hi, s1.hi, s2.hi, s3.hi};
        *reinterpret_cast<bf16x4*>(c.bl + o) = bf16x4{s0.lo, s1.lo, s2.lo, s3.lo};
    }
};

// OT: type of the optional operand copy (the 16-bit type in the fp32 split form whose backward runs on 16-bit operands: GemmArgs::save16)
// STATS: LayerNorm statistics of the row just produced, for the GEMM that consumes LN(row) in the folded form (GemmArgs::ln_part).  The
// epilogue sweep gives 16 consecutive lanes 64 consecutive columns of one row (ch = tid % (BN / 4), BN / 4 a multiple of 16), so a
// group's (sum, sum of squares about its own mean) is two 4-step DPP reductions inside the lane row -- no LDS, no atomics, fixed order
template <class AT, class OT = AT, bool STATS = false>
struct EpiBiasResid {
    const float* bias; const float* resid; float* out; OT* out_at; int ld;
    float2* part = nullptr;
    const float* rs = nullptr;   // stochastic depth (GemmArgs::row_scale): the branch (acc + bias) of image row / 197 is multiplied by rs[image]
    typedef Bias4 Col; typedef Raw4<float> Pre;
    __device__ __forceinline__ Col col_init(int col) const { return load_bias4(bias, col); }
    __device__ __forceinline__ Pre pre(int row, int col) const { return load_raw4(resid + (size_t)row * ld + col); }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre& p) const {
        const size_t o = (size_t)row * ld + col;
        float r[4];
        p.get(r);
        float v0 = a[0] + c.b[0] + r[0], v1 = a[1] + c.b[1] + r[1];
        float v2 = a[2] + c.b[2] + r[2], v3 = a[3] + c.b[3] + r[3];
        if (rs) {   // (uniform branch; off the default path)
            const float sc = rs[row / NT];
            v0 = fmaf(sc, a[0] + c.b[0], r[0]); v1 = fmaf(sc, a[1] + c.b[1], r[1]);
            v2 = fmaf(sc, a[2] + c.b[2], r[2]); v3 = fmaf(sc, a[3] + c.b[3], r[3]);
        }
        store4(out + o, v0, v1, v2, v3);
        if (out_at) store4(out_at + o, v0, v1, v2, v3);
        if constexpr (STATS) {
            const float sum = row16_sum((v0 + v1) + (v2 + v3));
            const float m = sum * (1.0f / 64.0f);
            const float d0 = v0 - m, d1 = v1 - m, d2 = v2 - m, d3 = v3 - m;
            const float m2 = row16_sum(fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3));
            if ((col & 63) == 0) part[(size_t)row * LN_PARTS + (col >> 6)] = make_float2(sum, m2);
        }
    }
};

// FAST (fp32 functor of the split forms): Phi by Abramowitz-Stegun 26.2.17 (the 16-bit functor's: |h err| 4e-7, |gelu' err| 3e-7 absolute,
// one v_exp + one v_rcp per element, packed fp32) instead of erff + expf -- below the split GEMMs' own 1e-6 and a tenth of the VALU work
// LNF: LayerNorm folded in (GemmArgs::ln_st): the accumulator is u16 . (gamma W)^T of the un-normalised row; z = rstd (acc - mean cs) + bias'
template <class AT, bool HAS_GP, class GT = AT, bool FAST = false, bool LNF = false>   // GT: type gelu'(z) is saved in (GemmArgs::save16: 16 bits under fp32 arithmetic)
struct EpiFc1 {
    const float* bias; AT* h; GT* gp; int ld;   // gp: gelu'(z), kept for the backward pass (training only)
    bf16* h3;   // split fp32 form: h goes out as the 16-bit hi / hi / lo operand of the fc2 GEMM ([rows, 3 ld]) instead of as fp32
    int f8 = 0; // ... in the hi16 / fp8 form (store4_split_f8)
    // LNF: column sums of W'; (mean, rstd) per LOGICAL row from ln_st (tile shapes without a statistics prologue: a pre-pass filled it), or --
    // RowStats kernels (gemm_bpre.h) -- merged from the producer's partials ln_part in the kernel prologue, which also leaves them in st_out
    const float* ln_cs = nullptr; const float2* ln_st = nullptr; const float2* ln_part = nullptr; float2* st_out = nullptr;
    struct ColF { float b[4]; float cs[4]; };
    typedef typename std::conditional<LNF, ColF, Bias4>::type Col;
    typedef typename std::conditional<LNF, float2, NoCtx>::type Pre;
    __device__ __forceinline__ Col col_init(int col) const {
        if constexpr (LNF) {
            ColF c;
            const Bias4 b4 = load_bias4(bias, col), c4 = load_bias4(ln_cs, col);
#pragma unroll
            for (int i = 0; i < 4; ++i) { c.b[i] = b4.b[i]; c.cs[i] = c4.b[i]; }
            return c;
        } else {
            return load_bias4(bias, col);
        }
    }
    __device__ __forceinline__ Pre pre(int row, int) const {
        if constexpr (LNF) return ln_st[row];   // (RowStats kernels do not call this)
        else return {};
    }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre& p) const {
        const size_t o = (size_t)row * ld + col;
        float hv[4], gv[4], z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (LNF) z[i] = fmaf(p.y, fmaf(-p.x, c.cs[i], a[i]), c.b[i]);
            else z[i] = a[i] + c.b[i];
        }
        if (HAS_GP) {
            if constexpr (sizeof(AT) == 2 || FAST) {   // two elements per packed-fp32 issue slot
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f32x2 h2, g2;
                    gelu_both_x2(f32x2{z[i], z[i + 1]}, h2, g2);
                    hv[i] = h2[0]; hv[i + 1] = h2[1]; gv[i] = g2[0]; gv[i + 1] = g2[1];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) gelu_both<AT>(z[i], hv[i], gv[i]);
            }
            store4_nt(gp + o, gv[0], gv[1], gv[2], gv[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = FAST ? gelu_fwd<bf16>(z[i]) : gelu_fwd<AT>(z[i]);
        }
        if constexpr (sizeof(AT) == 4) {
            if (h3) {
                if (f8) store4_split_f8(h3 + (size_t)row * SPLIT_A * ld, ld, col, hv[0], hv[1], hv[2], hv[3]);
                else store4_split3(h3 + (size_t)row * SPLIT_A * ld + col, ld, hv[0], hv[1], hv[2], hv[3]);
                return;
            }
        }
        store4(h + o, hv[0], hv[1], hv[2], hv[3]);
    }
};

// functors whose Pre is a per-row float2 that a kernel with a statistics prologue provides from LDS instead of calling pre()
template <class E> struct RowStats : std::false_type {};
template <class AT, bool G, class GT, bool F> struct RowStats<EpiFc1<AT, G, GT, F, true>> : std::true_type {};
// pre-pass for the other tile shapes: (mean, rstd) of logical row r = source row a_map[r] into st[r] and st_src[source row]
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ part, const int* __restrict__ a_map, float2* __restrict__ st,
                                                          float2* __restrict__ st_src, int rows) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int t = a_map ? a_map[r] : r;
    const float2 v = ln_merge_parts(part, t);
    st[r] = v;
    if (st_src) st_src[t] = v;
}

// PLAIN = no row map and no row mask (teacher pass / dense rows): the per-chunk context is then just the 16 B of the residual,
// small enough for the kernels to issue a whole pass of residual loads ahead of the staging barriers (PRE_ALL); with the
// 32-byte {dst, mask, residual} context of the general form the loads sit right in front of their use.
// `resid` is the residual the row is added to (x itself for the in-place form; the block's `u` when the adapter's
// up-projection rides along as an extra k-tile of the contraction (CatArgs): x = u + (h W2^T + b2) + (d_act (s Wup)^T + s b_up)
// is then ONE read of u and ONE write of x instead of two fp32 read-modify-write passes over [M,768]).
template <class AT, bool PLAIN, class HT = AT>   // HT: type the MLP output is saved in for the gate gradient (GemmArgs::save16)
struct EpiFc2 {
    const float* bias; float* x; const int* row_map; const float* row_mask; HT* h_out;
    const float* resid; const float* bias2; float scale2;
    const float* rs = nullptr;   // stochastic depth (GemmArgs::row_scale): the MLP branch of token row t is multiplied by rs[t / 197] (h_out keeps the unscaled value)
    // b: what is added to the accumulator for x (fc2 bias + s * up-projection bias); hb: the same for the saved MLP output h_out
    // (fc2 bias only: with the up-projection riding on the contraction h_out = mlp(x) + s up_nobias(d_act), and tok_bwd takes
    // <g, s up_nobias(d_act)> back out of the gate gradient -- without the bias term it needs no 768-wide dot for that)
    struct ColH { float b[4]; float hb[4]; };
    typedef typename std::conditional<PLAIN, Bias4, ColH>::type Col;
    struct PreG { int dst; float m; Raw4<float> r; };
    typedef typename std::conditional<PLAIN, Raw4<float>, PreG>::type Pre;
    __device__ __forceinline__ Col col_init(int col) const {
        Col c;
        const Bias4 c1 = load_bias4(bias, col);
#pragma unroll
        for (int i = 0; i < 4; ++i) c.b[i] = c1.b[i];
        if constexpr (!PLAIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c.hb[i] = c1.b[i];
        }
        if (bias2) {
            const Bias4 c2 = load_bias4(bias2, col);
#pragma unroll
            for (int i = 0; i < 4; ++i) c.b[i] = fmaf(scale2, c2.b[i], c.b[i]);
        }
        return c;
    }
    __device__ __forceinline__ Pre pre(int row, int col) const {
        if constexpr (PLAIN) {
            return load_raw4(resid + (size_t)row * D + col);
        } else {
            Pre p;
            p.dst = row_map ? row_map[row] : row;
            p.m = row_mask ? row_mask[p.dst] : 1.0f;
            p.r = load_raw4(resid + (size_t)p.dst * D + col);   // in place when resid == x: this chunk is the only writer of these 4 values
            return p;
        }
    }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre& p) const {
        const float h0 = a[0] + c.b[0], h1 = a[1] + c.b[1], h2 = a[2] + c.b[2], h3 = a[3] + c.b[3];
        float r[4];
        if constexpr (PLAIN) {
            if (h_out) store4_nt(h_out + (size_t)row * D + col, h0, h1, h2, h3);   // read again only by the backward pass
            p.get(r);
            const float sc = rs ? rs[row / NT] : 1.0f;
            store4(x + (size_t)row * D + col, fmaf(sc, h0, r[0]), fmaf(sc, h1, r[1]), fmaf(sc, h2, r[2]), fmaf(sc, h3, r[3]));
        } else {
            if (h_out) store4_nt(h_out + (size_t)row * D + col, a[0] + c.hb[0], a[1] + c.hb[1], a[2] + c.hb[2], a[3] + c.hb[3]);
            p.r.get(r);
            const float m = rs ? p.m * rs[p.dst / NT] : p.m;
            store4(x + (size_t)p.dst * D + col, r[0] + m * h0, r[1] + m * h1, r[2] + m * h2, r[3] + m * h3);
        }
    }
};

// MAPPED: gp is indexed by TOKEN row (dense "masked" forward) while this GEMM runs on compact rows.  A separate
// instantiation: the extra index load in front of every gelu' load costs the unmapped kernel 37 % (142 -> 195 us at B=128)
template <class AT, bool MAPPED>
struct EpiGeluBwd {
    const AT* gp; AT* out; int ld;   // gp = gelu'(z) saved by the fc1 epilogue
    const int* row_map;
    bf16* out3; float s3;   // split fp32 form: dZ * s3 goes out as the split operand of the fc1 dgrad GEMM instead of as fp32
    bool hi_only;           // ... its hi half alone (that GEMM contracts one part)
    typedef NoCtx Col; typedef Raw4<AT> Pre;
    __device__ __forceinline__ Col col_init(int) const { return {}; }
    __device__ __forceinline__ Pre pre(int row, int col) const {   // gelu'(z) is read exactly once: streaming load
        const size_t src = (size_t)(MAPPED ? row_map[row] : row) * ld + col;
        if constexpr (sizeof(AT) == 2) {
            Pre p;
            p.v = DYT_NT_LOAD(reinterpret_cast<const bf16x4*>(gp + src));
            return p;
        } else {
            return load_raw4(gp + src);
        }
    }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col&, const Pre& p) const {
        float g[4];
        p.get(g);
        if constexpr (sizeof(AT) == 4) {
            if (out3) { store4_split3(out3 + (size_t)row * SPLIT_A * ld + col, ld, a[0] * g[0] * s3, a[1] * g[1] * s3, a[2] * g[2] * s3, a[3] * g[3] * s3, hi_only); return; }
        }
        store4(out + (size_t)row * ld + col, a[0] * g[0], a[1] * g[1], a[2] * g[2], a[3] * g[3]);
    }
};

struct EpiStoreF32 {
    float* out; int ld; int accumulate; float alpha;   // out (+)= alpha * acc
    typedef NoCtx Col; typedef Raw4<float> Pre;
    __device__ __forceinline__ Col col_init(int) const { return {}; }
    __device__ __forceinline__ Pre pre(int row, int col) const {
        if (accumulate) return load_raw4(out + (size_t)row * ld + col);
        return {make_float4(0.f, 0.f, 0.f, 0.f)};
    }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col&, const Pre& p) const {
        float r[4];
        p.get(r);
        store4(out + (size_t)row * ld + col, r[0] + alpha * a[0], r[1] + alpha * a[1], r[2] + alpha * a[2], r[3] + alpha * a[3]);
    }
};

template <class AT>
struct EpiStoreAT {
    AT* out; int ld;
    typedef NoCtx Col; typedef NoCtx Pre;
    __device__ __forceinline__ Col col_init(int) const { return {}; }
    __device__ __forceinline__ Pre pre(int, int) const { return {}; }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col&, const Pre&) const {
        store4(out + (size_t)row * ld + col, a[0], a[1], a[2], a[3]);
    }
};

template <class AT>
struct EpiBiasAT {
    const float* bias; AT* out; int ld;
    typedef Bias4 Col; typedef NoCtx Pre;
    __device__ __forceinline__ Col col_init(int col) const { return load_bias4(bias, col); }
    __device__ __forceinline__ Pre pre(int, int) const { return {}; }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre&) const {
        store4(out + (size_t)row * ld + col, a[0] + c.b[0], a[1] + c.b[1], a[2] + c.b[2], a[3] + c.b[3]);
    }
};

template <class AT, class ST = AT>   // ST: type of the second copy (GemmArgs::save16: the 16-bit d_act the 16-bit backward reads)
struct EpiAdDown {
    const float* bias;  // padded to RP
    AT* out;            // [M, RP]
    const uint8_t* keep; int r; float inv_keep; float drop_p; uint64_t seed, subseq;
    const int* row_map;  // token row of compact row `row` (mask / RNG are indexed by token), or null
    const uint64_t* seed_dev;   // overrides `seed` when set (captured graphs draw fresh noise per replay)
    ST* out_s; float s_out;     // optional second copy s_out * result: the A2 operand of the fc2 + up-projection contraction (the adapter
                                // scale goes on the O(1) activations, not on the possibly tiny up-projection weights: fp16 subnormals)
    bf16* out3 = nullptr; float s3 = 1.f;   // split fp32 forms: s3 * result as a [rows][hi 64 | lo 64] image -- the up-projection's three-part operand
    typedef Bias4 Col;
    struct Pre { int trow; };
    __device__ __forceinline__ Col col_init(int col) const { return load_bias4(bias, col); }
    __device__ __forceinline__ Pre pre(int row, int) const { return {row_map ? row_map[row] : row}; }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre& p) const {
        const int trow = p.trow;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(a[i] + c.b[i], 0.0f);
        if (drop_p > 0.f) {
            if (keep) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = (col + i < r && keep[(size_t)trow * r + col + i]) ? v[i] * inv_keep : 0.0f;
            } else {
                Philox ph(seed_dev ? *seed_dev : seed, subseq, (uint64_t)trow * (RP / 4) + (col >> 2));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ph.u01(i) >= drop_p ? v[i] * inv_keep : 0.0f;
            }
        }
        store4(out + (size_t)row * RP + col, v[0], v[1], v[2], v[3]);
        if (out_s) store4(out_s + (size_t)row * RP + col, v[0] * s_out, v[1] * s_out, v[2] * s_out, v[3] * s_out);
        if constexpr (sizeof(AT) == 4) {
            if (out3) store4_split3(out3 + (size_t)row * (SPLIT_A * RP) + col, RP, v[0] * s3, v[1] * s3, v[2] * s3, v[3] * s3);
        }
    }
};

template <bool MAPPED>   // MAPPED: rows go through row_map (cls-only tail of the last block); see EpiFc2 for why two forms
struct EpiAdUp {
    const float* bias; const float* u; float* out; float scale; const int* row_map;
    const float* skip_mask;   // rows with skip_mask[row] != 0 are left alone (kept tokens: their fc2 launch adds the adapter itself)
    const float* rs = nullptr;   // MAPPED (cls-row proj of a complete_model pass's last block): stochastic-depth scale of image dst / 197
    float bscale = -1.f;         // >= 0: out = u + scale * acc + bscale * bias (GemmArgs::bias_scale: the A operand already carries the adapter scale)
    typedef Bias4 Col;
    struct PreG { int dst; Raw4<float> r; };
    typedef typename std::conditional<MAPPED, PreG, Raw4<float>>::type Pre;
    __device__ __forceinline__ Col col_init(int col) const { return load_bias4(bias, col); }
    __device__ __forceinline__ Pre pre(int row, int col) const {
        if constexpr (MAPPED) {
            Pre p;
            p.dst = row_map[row];
            p.r = load_raw4(u + (size_t)p.dst * D + col);
            return p;
        } else {
            if (skip_mask && skip_mask[row] != 0.f) return {make_float4(0.f, 0.f, 0.f, 0.f)};
            return load_raw4(u + (size_t)row * D + col);
        }
    }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre& p) const {
        float r[4];
        size_t dst = row;
        if constexpr (MAPPED) { p.r.get(r); dst = p.dst; } else { if (skip_mask && skip_mask[row] != 0.f) return; p.get(r); }
        float sc = scale;
        if constexpr (MAPPED) { if (rs) sc *= rs[dst / NT]; }
        if (bscale >= 0.f) {   // (uniform branch)
            store4(out + dst * D + col, r[0] + fmaf(bscale, c.b[0], sc * a[0]), r[1] + fmaf(bscale, c.b[1], sc * a[1]),
                   r[2] + fmaf(bscale, c.b[2], sc * a[2]), r[3] + fmaf(bscale, c.b[3], sc * a[3]));
            return;
        }
        store4(out + dst * D + col, r[0] + sc * (a[0] + c.b[0]), r[1] + sc * (a[1] + c.b[1]),
               r[2] + sc * (a[2] + c.b[2]), r[3] + sc * (a[3] + c.b[3]));
    }
};

template <class AT>
struct EpiAdDgradUp {
    const AT* dact; AT* out; float scale; float inv_keep;
    typedef NoCtx Col; typedef Raw4<AT> Pre;
    __device__ __forceinline__ Col col_init(int) const { return {}; }
    __device__ __forceinline__ Pre pre(int row, int col) const { return load_raw4(dact + (size_t)row * RP + col); }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col&, const Pre& p) const {
        float d[4];
        p.get(d);
        const float s = scale * inv_keep;
        store4(out + (size_t)row * RP + col, d[0] != 0.f ? a[0] * s : 0.f, d[1] != 0.f ? a[1] * s : 0.f,
               d[2] != 0.f ? a[2] * s : 0.f, d[3] != 0.f ? a[3] * s : 0.f);
    }
};

struct EpiEmbed {
    const float* bias; const float* pos; float* x0;
    typedef Bias4 Col; typedef Raw4<float> Pre;
    __device__ __forceinline__ Col col_init(int col) const { return load_bias4(bias, col); }
    __device__ __forceinline__ Pre pre(int row, int col) const {
        const int p = row % NP;
        return load_raw4(pos + (size_t)(1 + p) * D + col);
    }
    __device__ __forceinline__ void apply(int row, int col, const float (&a)[4], const Col& c, const Pre& pr) const {
        const int b = row / NP, p = row - b * NP;
        const size_t o = ((size_t)b * NT + 1 + p) * D + col;
        float ps[4];
        pr.get(ps);
        store4(x0 + o, a[0] + c.b[0] + ps[0], a[1] + c.b[1] + ps[1], a[2] + c.b[2] + ps[2], a[3] + c.b[3] + ps[3]);
    }
};

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------
// LDS-DMA (global_load_lds) completion is tracked by vmcnt.  hipcc usually drains it before a
// __syncthreads(), but NOT reliably (the software-pipelined loop below was compiled with only
// lgkmcnt(0) in front of s_barrier -> stale tiles at full occupancy).  Every barrier that publishes
// DMA'd data is therefore preceded by this explicit wait.
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// phase timers (measurement builds only, ABL == 9): cycles of prologue / main loop / epilogue summed over
// workgroups (wave 0 lane 0), [3] = workgroups counted
__device__ unsigned long long g_gemm_dbg[4];

// Tile BM x BN x 64, WAVES_M x WAVES_N waves of 64 lanes (each wave owns a (BM/WAVES_M) x (BN/WAVES_N)
// block of 16x16 MFMA tiles).  Shipped configurations:
//   128x128, 2x2 waves, 64 KB LDS, 2 workgroups / CU  -- N = 768 GEMMs (tile quantisation) and default
//   256x256, 2x4 waves, 128 KB LDS, 1 workgroup / CU  -- wide-N GEMMs: half the L2->LDS bytes per FLOP
//   128x64,  2x2 waves                                 -- adapter bottleneck (N = 64)
// ABL: 0 = product, 9 = the same kernel with the three phase timers (tools/gemm_bench.py)
// CAT: the contraction gets ONE extra leading k-tile taken from a second operand pair -- A2 [rows, 64] (rows optionally
// gathered through a2_map) against W2 [N, 64] -- i.e. C = A2 W2^T + A W^T in one accumulator chain (adapter up-projection
// riding on the fc2 GEMM: K = 64 + 3072).  The extra tile is stage 0 of the ring, so the main loop, its pointer
// registers and its schedule are the plain kernel's (the A / W pointers are pre-decremented by one tile).
struct CatArgs { const bf16* A2; const bf16* W2; const int* a2_map; float out_scale; int a_fold; int a_ld; int f8_begin; const int* w_exp; };
// LEAD (split fp32 forms, round 6): the second operand pair as THREE leading k-tiles of a three-part product -- A2 [rows][hi 64 | lo 64] (rows
// optionally gathered through a2_map), W2 [N][hi 64 | lo 64]: tiles (A2_hi, W2_hi), (A2_hi, W2_lo), (A2_lo, W2_hi) -- contracted into the
// accumulators by a small synchronous loop BEFORE the pipelined main loop starts (stage, wait, barrier, 2 x TM x TN MFMAs, barrier), so the
// main loop -- three-part fold form or fp8-correction form -- its registers and its schedule stay exactly the plain kernel's.  Three exposed
// tile latencies per workgroup against the fp32 read-modify-write launch of [M,768] this replaces (adapter up-projection riding on fc2).   // f8_begin / w_exp: F8 kernels -- first fp8 k-tile, device word with the weight image's exponent   // a_ld: row stride of split operands when the contraction runs over fewer than three parts (0: K - a_fold * 64)
// a_fold: k-tiles of ONE part of split operands stored [hi | lo] (row stride K - a_fold * 64): k-tile kt reads A column tile kt - (kt >= a_fold ? a_fold : 0) and W column tile kt - (kt >= 2 a_fold ? 2 a_fold : 0), i.e. [A_hi | A_hi | A_lo] x [W_hi | W_lo | W_hi]; 0 = plain   // out_scale: accumulators x this before the epilogue functor (split fp32 form; 1 elsewhere)
// One workgroup = one tile: `bid` of `nwg` logical workgroups that tile rows [m_begin, M).  A device function so that one launch
// can hold workgroups of two tile shapes (gemm_bf16_rows_kernel below).
// F8 ("fp16f8" form of the split contraction, dyt_common.h: store4_split_f8): both operand images are [hi16 | 2K bytes of fp8]; k-tiles
// [0, f8_begin) are f16 tiles of 64, the tiles after them fp8 tiles of 128 (the same 128 B per row and stage, the same fragment reads),
// multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 -- ONE instruction per fragment pair and k-tile instead of two, at the time of one
// f16 MFMA per 64 k: the lane's 32 operand bytes are its two 16-B fragment chunks {g, 4 + g} of the row (the same k subset on both
// sides, so the order inside the tile does not matter).  The E8M0 scale operand takes the images' powers of two back out:
// 2^-(ew+11) for the A_hi8 x W_lo8 tiles (first half), 2^-(ew+12) for the A_lo8 x W_hi8 tiles.
template <int BM, int BN, int WAVES_M, int WAVES_N, class Epi, int ABL, bool CAT, bool F8 = false, int LEAD = 0>
__device__ __forceinline__ void gemm_bf16_nt_tile(
    const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K, const int* __restrict__ m_dev,
    const int* __restrict__ a_map, int m_begin, const Epi& epi, const CatArgs& cat, int bid, int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // declared HERE, not passed in: a generic char* would lose the LDS address space
    constexpr int BK = 64;
    constexpr int NW = WAVES_M * WAVES_N, NTHR = 64 * NW;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // 1 KiB (8 rows x 128 B) per wave-instruction
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && WM % 16 == 0 && WN % 16 == 0, "tile/wave layout");
    constexpr bool BOTH_KS = (TM + TN) * 2 * 4 <= 72;  // hold both k-substeps' fragments when registers allow
    static_assert(!(F8 && CAT), "the fp8-correction form has no leading k-tile variant");
    static_assert(!(LEAD && CAT) && (LEAD == 0 || LEAD == 3), "LEAD: three leading tiles of a three-part product, split forms only");
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v8i __attribute__((ext_vector_type(8)));

    const int Mv = m_dev ? min(*m_dev, M) : M;
    // XCD-aware block remap (bijective): consecutive logical tiles share an A row panel and
    // should land on the same XCD's L2; hardware places block b on XCD b % 8.
    const int tiles_n = N / BN;
    // ... over the tiles that exist (device-side row count: see gemm_bpre.h), so that a compacted launch spreads over all eight XCDs
    const int nwg_v = min(nwg, ((max(Mv - m_begin, 0) + BM - 1) / BM) * tiles_n);
    if (bid >= nwg_v) return;
    const int q8 = nwg_v >> 3, r8 = nwg_v & 7, xcd = bid & 7, idx = bid >> 3;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
    const int m0 = m_begin + tm * BM, n0 = tn * BN;
    if (m0 >= Mv) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    unsigned long long t_start = 0, t_loop0 = 0, t_loop1 = 0;
    if (ABL == 9) t_start = __builtin_readcyclecounter();

    // ---- staging: lane -> (row-in-8, 16B slot); source chunk = slot ^ row ----
    const int lrow = lane >> 3, slot = lane & 7, chunk = slot ^ lrow;
    const bf16* a_src[A_INSTR];
    const bf16* b_src[B_INSTR];
    const int fold = cat.a_fold, lda = cat.a_ld ? cat.a_ld : K - fold * BK;   // split A operand stored [hi | lo]: the hi part serves the first two thirds of the contraction
    // F8 kernels: 32-bit byte offsets from the (uniform) operand bases instead of per-lane 64-bit pointers -- 8 VGPRs less where the
    // accumulators and two fragment generations already fill the register file (operand images stay below 4 GB: M x 4K bytes)
    unsigned a_o32[A_INSTR], b_o32[B_INSTR];
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lane_o32 = (unsigned)(lrow * lda + chunk * 8) * 2u;
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) {
        const int row = (t * NW + wave) * 8 + lrow;
        int grow = min(m0 + row, Mv - 1);
        if (a_map) grow = a_map[grow];  // gathered A rows (compacted MLP backward)
        a_src[t] = A + (size_t)grow * lda + chunk * 8 - (CAT ? BK : 0);
        a_o32[t] = (unsigned)(((size_t)grow * lda + chunk * 8) * 2);
    }
#pragma unroll
    for (int t = 0; t < B_INSTR; ++t) {
        const int row = (t * NW + wave) * 8 + lrow;
        b_src[t] = W + (size_t)(n0 + row) * lda + chunk * 8 - (CAT ? BK : 0);   // split form: W stored [hi | lo] like A (same row stride)
        b_o32[t] = (unsigned)(((size_t)(n0 + row) * lda + chunk * 8) * 2);
    }
    // one 1-KiB DMA piece (idx < A_INSTR: A rows, else W rows) -- issued interleaved with the MFMAs so the
    // in-order wave never sits behind a burst of LDS-DMA issues (each costs ~100+ cycles back-to-back)
    auto stage_one = [&](int buf, int kt, int idx) {
        char* base = smem + buf * STAGE;
        if constexpr (F8) {   // rows are read straight through: tile kt = bytes [128 kt, 128 kt + 128) of the row image
            if constexpr (!BOTH_KS) {
                // 256x256 kernel: everything but the lane's place inside a piece (row lrow of 8, 16-B chunk) is wave-uniform and stays
                // in scalar registers -- ONE vector register of addressing next to 128 accumulators and two fragment generations
                // (per-lane row pointers spilled, and a spill reload inside the loop waits for vmcnt(0), i.e. for the DMA in flight).
                // Hence no row gather and no clamp to the valid row count here: rows past it are read and never stored (the split
                // operand images are padded to whole 256-row tiles; run_f8 sends gathered launches to the 128x128 kernel).
                const bool isA = idx < A_INSTR;
                const int piece = isA ? idx : idx - A_INSTR;
                const size_t row0 = (size_t)((isA ? m0 : n0) + (piece * NW + wave_s) * 8);
                const char* g = reinterpret_cast<const char*>(isA ? A : W) + (row0 * (size_t)lda) * 2 + (size_t)kt * 128;
                unsigned o = lane_o32;
                asm volatile("" : "+v"(o));   // keeps the zero-extension in this block: "SGPR base + 32-bit VGPR offset" form of the load
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o),
                                                 (__attribute__((address_space(3))) void*)(base + (isA ? 0 : A_BYTES) + (piece * NW + wave_s) * 1024), 16, 0, 0);
                return;
            }
            const char* ga = reinterpret_cast<const char*>(A) + (size_t)kt * 128;
            const char* gw = reinterpret_cast<const char*>(W) + (size_t)kt * 128;
            unsigned o = idx < A_INSTR ? a_o32[idx] : b_o32[idx - A_INSTR];
            asm volatile("" : "+v"(o));
            if (idx < A_INSTR)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + o),
                                                 (__attribute__((address_space(3))) void*)(base + (idx * NW + wave) * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + o),
                                                 (__attribute__((address_space(3))) void*)(base + A_BYTES + ((idx - A_INSTR) * NW + wave) * 1024), 16, 0, 0);
            return;
        }
        if (idx < A_INSTR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[idx] + (kt - (kt >= fold ? fold : 0)) * BK),
                                             (__attribute__((address_space(3))) void*)(base + (idx * NW + wave) * 1024),
                                             16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(b_src[idx - A_INSTR] + (kt - (kt >= 2 * fold ? 2 * fold : 0)) * BK),
                (__attribute__((address_space(3))) void*)(base + A_BYTES + ((idx - A_INSTR) * NW + wave) * 1024), 16, 0, 0);
    };
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        if constexpr (F8) {
#pragma unroll
            for (int t = 0; t < A_INSTR + B_INSTR; ++t) stage_one(buf, kt, t);
            return;
        }
#pragma unroll
        for (int t = 0; t < A_INSTR; ++t)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[t] + (kt - (kt >= fold ? fold : 0)) * BK),
                                             (__attribute__((address_space(3))) void*)(base + (t * NW + wave) * 1024),
                                             16, 0, 0);
#pragma unroll
        for (int t = 0; t < B_INSTR; ++t)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[t] + (kt - (kt >= 2 * fold ? 2 * fold : 0)) * BK),
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + (t * NW + wave) * 1024),
                                             16, 0, 0);
    };

    // CAT: stage 0 comes from the second operand pair (one 64-wide row = one 128-B ring row)
    auto stage_first = [&]() {
        if constexpr (CAT) {
#pragma unroll
            for (int t = 0; t < A_INSTR; ++t) {
                int grow = min(m0 + (t * NW + wave) * 8 + lrow, Mv - 1);
                if (cat.a2_map) grow = cat.a2_map[grow];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cat.A2 + (size_t)grow * BK + chunk * 8),
                                                 (__attribute__((address_space(3))) void*)(smem + (t * NW + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < B_INSTR; ++t) {
                const int row = (t * NW + wave) * 8 + lrow;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cat.W2 + (size_t)(n0 + row) * BK + chunk * 8),
                                                 (__attribute__((address_space(3))) void*)(smem + A_BYTES + (t * NW + wave) * 1024), 16, 0, 0);
            }
        } else {
            stage(0, 0);
        }
    };

    // ---- fragment read offsets: row = tilebase + (lane & 15), chunk = ks*4 + (lane >> 4) ----
    const int frow = lane & 15;
    const int fslot0 = ((lane >> 4)) ^ (lane & 7);
    const int fslot1 = (4 + (lane >> 4)) ^ (lane & 7);
    const int a_off = (wm * WM + frow) * 128;
    const int b_off = A_BYTES + (wn * WN + frow) * 128;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (LEAD > 0) {
        // ---- leading tiles of the second operand pair (see CatArgs).  The four operand tiles go out in ONE DMA round -- A2_hi / W2_hi into
        // ring slot 0, A2_lo / W2_lo into slot 1 -- so the three products pay one exposed load latency and two barriers:
        // (A2_hi, W2_hi), (A2_hi, W2_lo), (A2_lo, W2_hi)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int q = 0; q < A_INSTR; ++q) {
                int grow = min(m0 + (q * NW + wave) * 8 + lrow, Mv - 1);
                if (cat.a2_map) grow = cat.a2_map[grow];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cat.A2 + (size_t)grow * (2 * BK) + part * BK + chunk * 8),
                                                 (__attribute__((address_space(3))) void*)(smem + part * STAGE + (q * NW + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < B_INSTR; ++q) {
                const int row = (q * NW + wave) * 8 + lrow;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cat.W2 + (size_t)(n0 + row) * (2 * BK) + part * BK + chunk * 8),
                                                 (__attribute__((address_space(3))) void*)(smem + part * STAGE + A_BYTES + (q * NW + wave) * 1024), 16, 0, 0);
            }
        }
        dma_wait_all();
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < LEAD; ++t) {
            const char* ab = smem + (t == 2 ? STAGE : 0);
            const char* wb = smem + (t == 1 ? STAGE : 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = (ks == 0 ? fslot0 : fslot1) * 16;
                bf16x8 la[TM], lw[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) la[i] = *reinterpret_cast<const bf16x8*>(ab + a_off + i * 2048 + so);
#pragma unroll
                for (int j = 0; j < TN; ++j) lw[j] = *reinterpret_cast<const bf16x8*>(wb + b_off + j * 2048 + so);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = DYT_MFMA_16x16x32(lw[j], la[i], acc[i][j]);
            }
        }
        __syncthreads();   // every wave's reads of the two slots have returned before the main loop's first stages land in them
    }
    const int nk = K / BK + (CAT ? 1 : 0);
    if (ABL == 9) t_loop0 = __builtin_readcyclecounter();
    // fp8 tiles: E8M0 scale operands of the two correction products (the W fragment is the MFMA's A operand: the whole factor goes there)
    int f8_sc1 = 127, f8_sc2 = 127, f8_half = 0;
    if constexpr (F8) {
        const int ew = cat.w_exp ? *cat.w_exp : 0;
        f8_sc1 = 127 - (ew + 11); f8_sc2 = 127 - (ew + F8_LO_LOG2);
        f8_half = cat.f8_begin + (nk - cat.f8_begin) / 2;
    }
#define DYT_MFMA_F8(w8, a8, c, sc) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4((w8), (a8), (c), 0, 0, 0, (sc), 0, 127)
#define DYT_LO4(x) __builtin_shufflevector((x), (x), 0, 1, 2, 3)
#define DYT_HI4(x) __builtin_shufflevector((x), (x), 4, 5, 6, 7)
#define DYT_CAT8(lo, hi) __builtin_shufflevector((lo), (hi), 0, 1, 2, 3, 4, 5, 6, 7)
    if constexpr (!BOTH_KS && F8) {
        // ---- the half-stage pipeline below with 8-register fragments: .lo = k-substep 0 (row chunk g of lane group g), .hi = k-substep 1
        // (chunk 4 + g).  f16 tiles run exactly the plain schedule on the halves.  An fp8 tile needs BOTH halves per MFMA, so its 32
        // MFMAs are split by ROW fragments instead: block 1 = rows 0..3 (while rows 4..7 of the tile are read), barrier, block 2 =
        // rows 4..7, j-major (while rows 0..3 of the NEXT tile, then -- each after its last use -- its W fragments are read and the
        // DMA of the tile after that goes out).  Same reads, DMA pieces and matrix-pipe cycles per k-tile as an f16 tile.
        bf16x8 faA[TM], fwA[TN], faB[TM], fwB[TN];   // f16 tiles: the plain kernel's two fragment sets
        constexpr int NDMA = A_INSTR + B_INSTR;
        static_assert(TM == 8 && TN == 4 && NDMA == 8, "interleave pattern written for the 128x64 wave tile, 8 DMA pieces");
#define DYT_LDS4(off) (*reinterpret_cast<const v4i*>(base + (off)))
        stage_first();
        stage(1, 1);
        dma_wait_all();
        __syncthreads();
        {
            const char* base = smem;
#pragma unroll
            for (int i = 0; i < TM; ++i) faA[i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 2048 + fslot0 * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) fwA[j] = *reinterpret_cast<const bf16x8*>(base + b_off + j * 2048 + fslot0 * 16);
        }
#define DYT_MMA(FW, FA)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)            \
        acc[i][j] = DYT_MFMA_16x16x32(FW[j], FA[i], acc[i][j]);
#define DYT_SG3R __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#define DYT_SG2R __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#define DYT_SGB /* per 8 MFMAs: M2 R M2 R M1 D M2 R M1 D */                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1);
        const int n1 = cat.f8_begin;
        for (int kt = 0; kt < n1; ++kt) {   // f16 tiles: the plain kernel's iteration (a tile kt + 1 always exists: the fp8 tiles follow)
            {
                const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
                for (int i = 0; i < TM; ++i) faB[i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 2048 + fslot1 * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) fwB[j] = *reinterpret_cast<const bf16x8*>(base + b_off + j * 2048 + fslot1 * 16);
            }
            DYT_MMA(fwA, faA)
            DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R
            DYT_SG2R DYT_SG2R DYT_SG2R DYT_SG2R
            __builtin_amdgcn_sched_barrier(0);
            dma_wait_all();
            __syncthreads();
            // block B in issue order, fenced (next to the opaque staging offsets the group barriers no longer reproduce the plain
            // kernel's interleave): per 8 MFMAs  M2 R M2 R M1 D M2 R M1 D
            const int nxt = min(kt + 2, nk - 1);
            {
                const char* base = smem + ((kt + 1) & 1) * STAGE;
                const int so = fslot0 * 16;
#define DYT_RA(i) faA[i] = *reinterpret_cast<const bf16x8*>(base + a_off + (i) * 2048 + so);
#define DYT_RW(j) fwA[j] = *reinterpret_cast<const bf16x8*>(base + b_off + (j) * 2048 + so);
#define DYT_MB(i, j) acc[i][j] = DYT_MFMA_16x16x32(fwB[j], faB[i], acc[i][j]);
#define DYT_FN __builtin_amdgcn_sched_barrier(0);
#define DYT_GRP(g, R0, R1, R2)                                                                     \
                DYT_MB(2 * g, 0) DYT_MB(2 * g, 1) R0 DYT_FN DYT_MB(2 * g, 2) DYT_MB(2 * g, 3) R1 DYT_FN \
                DYT_MB(2 * g + 1, 0) stage_one(kt & 1, nxt, 2 * g); DYT_FN                               \
                DYT_MB(2 * g + 1, 1) DYT_MB(2 * g + 1, 2) R2 DYT_FN DYT_MB(2 * g + 1, 3) stage_one(kt & 1, nxt, 2 * g + 1); DYT_FN
                DYT_GRP(0, DYT_RA(0), DYT_RA(1), DYT_RA(2))
                DYT_GRP(1, DYT_RA(3), DYT_RA(4), DYT_RA(5))
                DYT_GRP(2, DYT_RA(6), DYT_RA(7), DYT_RW(0))
                DYT_GRP(3, DYT_RW(1), DYT_RW(2), DYT_RW(3))
#undef DYT_GRP
#undef DYT_FN
#undef DYT_MB
#undef DYT_RA
#undef DYT_RW
            }
        }
#undef DYT_MMA
        // fp8 tiles: 8-register fragments (.lo = the k-substep-0 chunk, .hi = the k-substep-1 chunk); they start with rows 0..3 and W of
        // tile n1 (the f16 loop's last prefetch into its own set is not reused: one exposed LDS round trip per GEMM tile)
        v8i fa[TM], fw[TN];
        {
            const char* base = smem + (n1 & 1) * STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = DYT_CAT8(DYT_LDS4(a_off + i * 2048 + fslot0 * 16), DYT_LDS4(a_off + i * 2048 + fslot1 * 16));
#pragma unroll
            for (int j = 0; j < TN; ++j) fw[j] = DYT_CAT8(DYT_LDS4(b_off + j * 2048 + fslot0 * 16), DYT_LDS4(b_off + j * 2048 + fslot1 * 16));
        }
        // The scheduler's group barriers do not place the scaled MFMAs (the schedule came out as all reads first, then the MFMAs, with
        // ~250 spilled registers), so the fp8 blocks are written in issue order and fenced with sched_barrier(0) every four MFMAs.
#define DYT_M8(i, j) acc[i][j] = DYT_MFMA_F8(fw[j], fa[i], acc[i][j], sc);
#define DYT_FENCE __builtin_amdgcn_sched_barrier(0);
#define DYT_RA8(i) fa[i] = DYT_CAT8(DYT_LDS4(a_off + (i) * 2048 + fslot0 * 16), DYT_LDS4(a_off + (i) * 2048 + fslot1 * 16));
#define DYT_RW8(j) fw[j] = DYT_CAT8(DYT_LDS4(b_off + (j) * 2048 + fslot0 * 16), DYT_LDS4(b_off + (j) * 2048 + fslot1 * 16));
        // block 1: rows 0..3 x all W fragments (16 MFMAs); both halves of rows 4..7 of the same tile are read meanwhile
#define DYT_F8_BLOCK1(kt_)                                                        \
        {                                                                         \
            const char* base = smem + ((kt_) & 1) * STAGE;                        \
            DYT_M8(0, 0) DYT_M8(0, 1) DYT_RA8(4) DYT_M8(0, 2) DYT_M8(0, 3) DYT_FENCE \
            DYT_M8(1, 0) DYT_M8(1, 1) DYT_RA8(5) DYT_M8(1, 2) DYT_M8(1, 3) DYT_FENCE \
            DYT_M8(2, 0) DYT_M8(2, 1) DYT_RA8(6) DYT_M8(2, 2) DYT_M8(2, 3) DYT_FENCE \
            DYT_M8(3, 0) DYT_M8(3, 1) DYT_RA8(7) DYT_M8(3, 2) DYT_M8(3, 3) DYT_FENCE \
        }
        DYT_FENCE
        for (int kt = n1; kt < nk - 1; ++kt) {
            const int sc = kt < f8_half ? f8_sc1 : f8_sc2;
            DYT_F8_BLOCK1(kt)
            dma_wait_all();   // this wave's pieces of stage kt+1 have landed ...
            __syncthreads();  // ... everywhere; every wave's reads of slot kt&1 have returned
            // block 2: rows 4..7, j-major (16 MFMAs); rows 0..3 of tile kt+1, its W fragments (each after the four MFMAs that read
            // the old one) and the 8 DMA pieces of stage kt+2 -> slot kt&1
            const int nxt = min(kt + 2, nk - 1);
            {
                const char* base = smem + ((kt + 1) & 1) * STAGE;
                DYT_M8(4, 0) DYT_M8(5, 0) DYT_RA8(0) stage_one(kt & 1, nxt, 0); DYT_FENCE
                DYT_M8(6, 0) DYT_M8(7, 0) stage_one(kt & 1, nxt, 1); DYT_FENCE
                DYT_M8(4, 1) DYT_M8(5, 1) DYT_RA8(1) stage_one(kt & 1, nxt, 2); DYT_FENCE
                DYT_M8(6, 1) DYT_M8(7, 1) DYT_RW8(0) stage_one(kt & 1, nxt, 3); DYT_FENCE
                DYT_M8(4, 2) DYT_M8(5, 2) DYT_RA8(2) stage_one(kt & 1, nxt, 4); DYT_FENCE
                DYT_M8(6, 2) DYT_M8(7, 2) DYT_RW8(1) stage_one(kt & 1, nxt, 5); DYT_FENCE
                DYT_M8(4, 3) DYT_M8(5, 3) DYT_RA8(3) stage_one(kt & 1, nxt, 6); DYT_FENCE
                DYT_M8(6, 3) DYT_M8(7, 3) DYT_RW8(2) stage_one(kt & 1, nxt, 7); DYT_FENCE
                DYT_RW8(3) DYT_FENCE
            }
        }
        {   // last tile: nothing left to prefetch
            const int kt = nk - 1;
            const int sc = kt < f8_half ? f8_sc1 : f8_sc2;
            DYT_F8_BLOCK1(kt)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 4; i < TM; ++i) acc[i][j] = DYT_MFMA_F8(fw[j], fa[i], acc[i][j], sc);
        }
#undef DYT_M8
#undef DYT_FENCE
#undef DYT_RA8
#undef DYT_RW8
#undef DYT_F8_BLOCK1
#undef DYT_SG3R
#undef DYT_SG2R
#undef DYT_SGB
#undef DYT_LDS4
    } else if constexpr (!BOTH_KS) {
        // ---- big wave tiles (128x64 per wave): half-stage software pipeline.  Two fragment sets, one per
        // k-substep; every LDS fragment read is issued one MFMA block (32 MFMAs) ahead of its use, the
        // single barrier of a stage sits between the two MFMA blocks, and the DMA of stage kt+2 is issued
        // right after it.  Invariant at the top of iteration kt: Fa holds (kt, ks=0); slot kt&1 holds
        // stage kt; slot (kt+1)&1 holds or is receiving stage kt+1.
        // (Round 5: an L2 prefetch of the stage 2-3 k-tiles ahead -- one global_load_dword per wave into a dead register behind the DMA pieces,
        // counted vmcnt(1) in front of the barrier -- was built to shorten the launch on operands that are not in the Infinity Cache (112 vs
        // 79 us, tools/gemm_bench.py COLD=1): cold launches unchanged, step 24.19 vs 24.01 ms same-box.  Not kept)
        bf16x8 faA[TM], fwA[TN], faB[TM], fwB[TN];
        auto read_a = [&](int buf) {   // (stage in `buf`, ks = 0) -> set A
            const char* base = smem + buf * STAGE;
#pragma unroll
            for (int i = 0; i < TM; ++i) faA[i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 2048 + fslot0 * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) fwA[j] = *reinterpret_cast<const bf16x8*>(base + b_off + j * 2048 + fslot0 * 16);
        };
        auto read_b = [&](int buf) {   // (stage in `buf`, ks = 1) -> set B
            const char* base = smem + buf * STAGE;
#pragma unroll
            for (int i = 0; i < TM; ++i) faB[i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 2048 + fslot1 * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) fwB[j] = *reinterpret_cast<const bf16x8*>(base + b_off + j * 2048 + fslot1 * 16);
        };
        stage_first();
        if (nk > 1) stage(1, 1);
        dma_wait_all();
        __syncthreads();
        read_a(0);
        constexpr int NDMA = A_INSTR + B_INSTR;
        static_assert(TM == 8 && TN == 4 && NDMA == 8, "interleave pattern written for the 128x64 wave tile, 8 DMA pieces");
#define DYT_MMA(FW, FA)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)            \
        acc[i][j] = DYT_MFMA_16x16x32(FW[j], FA[i], acc[i][j]);
#define DYT_SG3R __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#define DYT_SG2R __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#define DYT_SGB /* per 8 MFMAs: M2 R M2 R M1 D M2 R M1 D */                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1);
        for (int kt = 0; kt < nk - 1; ++kt) {
            // ---- block A: 32 MFMAs on set A, the 12 fragment reads of (kt, ks=1) -> set B slotted in between
            read_b(kt & 1);
            DYT_MMA(fwA, faA)
            DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R   // 24 MFMA, 8 reads
            DYT_SG2R DYT_SG2R DYT_SG2R DYT_SG2R                                       //  8 MFMA, 4 reads
            __builtin_amdgcn_sched_barrier(0);
            dma_wait_all();   // this wave's pieces of stage kt+1 have landed ...
            __syncthreads();  // ... everywhere; every wave's reads of slot kt&1 have returned
            // ---- block B: 32 MFMAs on set B; the reads of (kt+1, ks=0) -> set A and the 8 DMA pieces of stage
            //      kt+2 -> slot kt&1 slotted in between.  (Past the end the last stage is simply re-fetched
            //      into the free slot: keeps this a single basic block for the scheduler.)
            // source order = issue order (LDS-DMA writes and ds_reads are kept in program order by the
            // scheduler's memory dependencies): R R D R D, four times = 12 reads + 8 DMA pieces
            const int nxt = min(kt + 2, nk - 1);
            {
                const char* base = smem + ((kt + 1) & 1) * STAGE;
                const int so = fslot0 * 16;
#define DYT_RA(i) faA[i] = *reinterpret_cast<const bf16x8*>(base + a_off + (i) * 2048 + so);
#define DYT_RW(j) fwA[j] = *reinterpret_cast<const bf16x8*>(base + b_off + (j) * 2048 + so);
                DYT_RA(0) DYT_RA(1) stage_one(kt & 1, nxt, 0); DYT_RA(2) stage_one(kt & 1, nxt, 1);
                DYT_RA(3) DYT_RA(4) stage_one(kt & 1, nxt, 2); DYT_RA(5) stage_one(kt & 1, nxt, 3);
                DYT_RA(6) DYT_RA(7) stage_one(kt & 1, nxt, 4); DYT_RW(0) stage_one(kt & 1, nxt, 5);
                DYT_RW(1) DYT_RW(2) stage_one(kt & 1, nxt, 6); DYT_RW(3) stage_one(kt & 1, nxt, 7);
#undef DYT_RA
#undef DYT_RW
            }
            DYT_MMA(fwB, faB)
            DYT_SGB DYT_SGB DYT_SGB DYT_SGB
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // last stage: nothing left to prefetch
            read_b((nk - 1) & 1);
            DYT_MMA(fwA, faA)
            DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R DYT_SG3R
            DYT_SG2R DYT_SG2R DYT_SG2R DYT_SG2R
            __builtin_amdgcn_sched_barrier(0);
            DYT_MMA(fwB, faB)
        }
#undef DYT_MMA
#undef DYT_SG3R
#undef DYT_SG2R
#undef DYT_SGB
    } else {
    stage_first();
    for (int kt = 0; kt < nk; ++kt) {
        dma_wait_all();
        __syncthreads();  // stage kt landed (vmcnt drained before the barrier); ring slot (kt+1)&1 is free
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* base = smem + (kt & 1) * STAGE;
        // issue all fragment reads of the K step, then the MFMAs: one LDS-latency exposure per step
        bf16x8 af[2][TM], wf[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = (ks == 0 ? fslot0 : fslot1) * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 2048 + so);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(base + b_off + j * 2048 + so);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (F8 && kt >= cat.f8_begin) {   // fp8 tile: one MFMA per fragment pair over both k-substeps' chunks
            const int sc = kt < f8_half ? f8_sc1 : f8_sc2;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = DYT_MFMA_F8(DYT_CAT8(__builtin_bit_cast(v4i, wf[0][j]), __builtin_bit_cast(v4i, wf[1][j])),
                                            DYT_CAT8(__builtin_bit_cast(v4i, af[0][i]), __builtin_bit_cast(v4i, af[1][i])), acc[i][j], sc);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = DYT_MFMA_16x16x32(wf[ks][j], af[ks][i], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    }  // main-loop variants
#undef DYT_MFMA_F8
#undef DYT_LO4
#undef DYT_HI4
#undef DYT_CAT8
    if (ABL == 9) t_loop1 = __builtin_readcyclecounter();

    {
        // ---- epilogue through LDS: acc[i][j][e] = C[wm*WM + i*16 + (lane&15)][wn*WN + j*16 + (lane>>4)*4 + e].
        // The fragment layout gives each store instruction 16 rows x 64 B; instead the tile is parked in the
        // (now free) staging ring as fp32 with the 16-B chunk index XOR-swizzled by (row & 7) (conflict-free
        // ds_write_b128 / ds_read_b128) and read back row-major, so every global access of the epilogue
        // functor is a full-row, 16-B-per-lane coalesced transaction.  If the whole fp32 tile does not fit
        // the ring it goes through in WAVES_M passes of WM rows.
        constexpr bool ONE_PASS = BM * BN * 4 <= 2 * STAGE;
        constexpr int PASSES = ONE_PASS ? 1 : WAVES_M;
        constexpr int PROWS = BM / PASSES;
        static_assert(PROWS * BN * 4 <= 2 * STAGE, "epilogue staging does not fit the LDS ring");
        float* Cs = reinterpret_cast<float*>(smem);
        constexpr int CH = BN / 4;                 // 16-B chunks per tile row
        static_assert(NTHR % CH == 0, "a lane must keep one column chunk for the whole tile");
        constexpr int RSTEP = NTHR / CH;           // rows covered by one sweep of the workgroup
        constexpr int ITERS = PROWS / RSTEP;
        constexpr int BATCH = ITERS % 8 == 0 ? 8 : ITERS;
        const int ch = tid % CH, rl0 = tid / CH;
        const int col = n0 + ch * 4;
        const typename Epi::Col cc = epi.col_init(col);   // bias etc.: once per tile, not once per chunk
        dma_wait_all();  // nothing may still be landing in the ring when it is reused as staging
#pragma unroll 1
        for (int p = 0; p < PASSES; ++p) {
            // the functor's own global loads for the whole pass go out first: their latency overlaps the
            // staging write, the barriers and the LDS read-back instead of serialising chunk by chunk
            // (functors with a large per-chunk context prefetch one read-back batch at a time instead: 16 x 24 B
            // next to 128 live accumulators spilled)
            constexpr bool PRE_ALL = sizeof(typename Epi::Pre) * ITERS <= 256;
            typename Epi::Pre pr[PRE_ALL ? ITERS : BATCH];
            if (PRE_ALL) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it)
                    pr[it] = epi.pre(min(m0 + p * PROWS + rl0 + it * RSTEP, Mv - 1), col);
            }
            __syncthreads();
            if (ONE_PASS || wm == p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int rl = (ONE_PASS ? wm * WM : 0) + i * 16 + frow;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int chw = ((wn * WN + j * 16) >> 2) + (lane >> 4);
                        *reinterpret_cast<f32x4*>(Cs + rl * BN + ((chw ^ (rl & 7)) << 2)) = acc[i][j];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
                if (!PRE_ALL) {
#pragma unroll
                    for (int u = 0; u < BATCH; ++u)
                        pr[u] = epi.pre(min(m0 + p * PROWS + rl0 + (it0 + u) * RSTEP, Mv - 1), col);
                }
                f32x4 c4[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int rl = rl0 + (it0 + u) * RSTEP;
                    c4[u] = *reinterpret_cast<const f32x4*>(Cs + rl * BN + ((ch ^ (rl & 7)) << 2));
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int row = m0 + p * PROWS + rl0 + (it0 + u) * RSTEP;
                    if (row < Mv) {
                        const float os = cat.out_scale;   // 1.0 (exact) except in the split fp32 form of a gradient GEMM
                        const float v[4] = {c4[u][0] * os, c4[u][1] * os, c4[u][2] * os, c4[u][3] * os};
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

template <int BM, int BN, int WAVES_M, int WAVES_N, class Epi, int ABL = 0, bool CAT = false, bool F8 = false, int LEAD = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void gemm_bf16_nt_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K, const int* __restrict__ m_dev,
    const int* __restrict__ a_map, int m_begin, Epi epi, CatArgs cat) {
    // rows [m_begin, M) are tiled by this launch
    gemm_bf16_nt_tile<BM, BN, WAVES_M, WAVES_N, Epi, ABL, CAT, F8, LEAD>(A, W, M, N, K, m_dev, a_map, m_begin, epi, cat, blockIdx.x, gridDim.x);
}

// Narrow-N GEMM (N = 768) in ONE launch: the first n_big workgroups take 256x256 tiles of rows [0, body) -- whole rounds of the
// 256 CUs --, the others 128x128 tiles (eight waves as 2x4) of rows [body, M).  Both shapes accumulate every dot product in the
// same k order, so results do not depend on where the split falls; a compacted launch (device-side row count) leaves the
// workgroups beyond the count with nothing to do instead of a second, empty launch.
template <class Epi, bool CAT>
__global__ __launch_bounds__(512, 2) void gemm_bf16_rows_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K, const int* __restrict__ m_dev,
    const int* __restrict__ a_map, int body, int n_big, Epi epi, CatArgs cat) {
    const int b = blockIdx.x;
    if (b < n_big) gemm_bf16_nt_tile<256, 256, 2, 4, Epi, 0, CAT>(A, W, body, N, K, m_dev, a_map, 0, epi, cat, b, n_big);
    else gemm_bf16_nt_tile<128, 128, 2, 4, Epi, 0, CAT>(A, W, M, N, K, m_dev, a_map, body, epi, cat, b - n_big, (int)gridDim.x - n_big);
}

}  // namespace dyt
#include "gemm_bpre.h"
#include "gemm_f32_mfma.h"
#include "gemm_skinny.h"
namespace dyt {

// ------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------
// kernel launches of the bf16 family since the last reset (a logical GEMM may be two launches: bench.py converts
// the per-kernel-launch PMC traffic to its per-GEMM unit with this)
static long long g_bf16_kernel_launches = 0;
long long gemm_kernel_launch_count(int reset) {
    const long long n = g_bf16_kernel_launches;
    if (reset) g_bf16_kernel_launches = 0;
    return n;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int ABL, class Epi, bool CAT = false, bool F8 = false, int LEAD = 0>
static int launch_bf16_cfg(const GemmArgs& a, const Epi& epi, hipStream_t s, int m_begin = 0, int m_end = -1) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    if (m_end < 0) m_end = a.M;
    if (m_end <= m_begin) return 0;
    const int grid = ((m_end - m_begin + BM - 1) / BM) * (a.N / BN);
    const size_t lds = 2 * (BM + BN) * 64 * 2;
    auto kern = gemm_bf16_nt_kernel<BM, BN, WAVES_M, WAVES_N, Epi, ABL, CAT, F8, LEAD>;
    static bool attr_set[64] = {};   // per device: the attribute belongs to the (kernel, device) pair
    int dev = 0;
    DYT_HIP_CHECK(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        DYT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
        attr_set[dev & 63] = true;
    }
    const CatArgs cat{static_cast<const bf16*>(a.A2), static_cast<const bf16*>(a.W2), a.a2_map, a.out_scale, a.a_fold, a.a_ld, a.f8_begin, a.w_exp};
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, s, static_cast<const bf16*>(a.A), static_cast<const bf16*>(a.W),
                       m_end, a.N, a.K, a.m_dev, a.a_map, m_begin, epi, cat);
    ++g_bf16_kernel_launches;
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <class Epi, bool CAT>
static int launch_bf16_rows(const GemmArgs& a, const Epi& epi, hipStream_t s, int body) {
    const int n_big = (body / 256) * (a.N / 256), n_small = ((a.M - body + 127) / 128) * (a.N / 128);
    const size_t lds = 2 * (256 + 256) * 64 * 2;
    auto kern = gemm_bf16_rows_kernel<Epi, CAT>;
    static bool attr_set[64] = {};
    int dev = 0;
    DYT_HIP_CHECK(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        DYT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev & 63] = true;
    }
    const CatArgs cat{static_cast<const bf16*>(a.A2), static_cast<const bf16*>(a.W2), a.a2_map, a.out_scale, a.a_fold, a.a_ld, 0, nullptr};
    hipLaunchKernelGGL(kern, dim3(n_big + n_small), dim3(512), lds, s, static_cast<const bf16*>(a.A), static_cast<const bf16*>(a.W), a.M, a.N,
                       a.K, a.m_dev, a.a_map, body, n_big, epi, cat);
    ++g_bf16_kernel_launches;
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int ABL, class Epi>
static int launch_bf16_bpre(const GemmArgs& a, const Epi& epi, hipStream_t s, int m_begin = 0, int m_end = -1) {
    if (m_end < 0) m_end = a.M;
    if (m_end <= m_begin) return 0;
    if (a.K % 256 != 0 || a.N % 256 != 0) { set_error("gemm_bpre: K=%d and N=%d must be multiples of 256", a.K, a.N); return -1; }
    const int grid = ((m_end - m_begin + 127) / 128) * (a.N / 256);
    hipLaunchKernelGGL((gemm_bf16_bpre_kernel<Epi, ABL>), dim3(grid), dim3(256), 0, s, static_cast<const bf16*>(a.A),
                       static_cast<const bf16*>(a.W), m_end, a.N, a.K, a.m_dev, a.a_map, m_begin, epi, a.out_scale);
    ++g_bf16_kernel_launches;
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_preshuffle_w(const void* W, void* Wp, int N, int K, hipStream_t s) {
    if (N % 16 != 0 || K % 32 != 0) { set_error("preshuffle: N=%d %% 16, K=%d %% 32 required", N, K); return -1; }
    const size_t chunks = (size_t)N * K / 8;
    hipLaunchKernelGGL(preshuffle_w_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s,
                       static_cast<const bf16*>(W), static_cast<bf16*>(Wp), N, K);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

static int g_splitk = -1;   // DYT_OPT_GEMM_SPLITK (process-wide; environment DYT_SPLITK sets the default)
void set_gemm_splitk(int v) { g_splitk = v != 0; }
int get_gemm_splitk() {
    if (g_splitk < 0) { const char* e = getenv("DYT_SPLITK"); g_splitk = e ? (atoi(e) != 0) : 1; }
    return g_splitk;
}
static int g_big_tile_min_n = 2304;
static int g_use_bpre = 1;        // wide-N GEMMs with a pre-shuffled frozen weight: 128x256 tiles, 2 workgroups / CU (gemm_bpre.h)
static int g_split_rows = getenv("DYT_SPLIT_ROWS") ? atoi(getenv("DYT_SPLIT_ROWS")) : 1;   // (0: 128x128 tiles for every row -- measurement knob)
//      // narrow-N GEMMs: 256x256 tiles for whole rounds of rows + 128x128 tiles for the rest  // N >= this (and % 256 == 0): 256x256 tiles with the half-stage pipeline

// the "fp16f8" split contraction (a.K = 2 x the logical K: K / 64 f16 tiles + K / 64 fp8 tiles): tile shapes as for the three-part form
template <class Epi, int LEAD = 0>
static int run_f8(const GemmArgs& a, const Epi& epi, hipStream_t s) {
    if (a.K % 256 != 0 || a.M <= 0 || a.N % 128 != 0 || a.f8_begin * 128 != a.K) { set_error("gemm f8 form: K=%d N=%d M=%d", a.K, a.N, a.M); return -1; }
    if (LEAD && (!a.A2 || !a.W2)) { set_error("gemm f8 form: leading tiles need A2 and W2"); return -1; }
    if (a.N % 256 == 0 && a.N >= 2304 && a.M >= 2048 && !a.a_map) return launch_bf16_cfg<256, 256, 2, 4, 0, Epi, false, true, LEAD>(a, epi, s);
    // logical K <= 768, N = 768 (proj forward of the split modes): like the 16-bit modes' proj (run_bf16: shortk_n768) -- one launch of 128x128 tiles,
    // two workgroups per CU, instead of a 256x256 body + a 128x128 row tail.  DYT_F8_SHORTK_SMALL=0: the body + tail scheme
    static const int f8_shortk_small = getenv("DYT_F8_SHORTK_SMALL") ? atoi(getenv("DYT_F8_SHORTK_SMALL")) : 1;
    if (f8_shortk_small && (a.K <= 2 * D || f8_shortk_small == 2) && a.N % 128 == 0 && a.N < 2304) return launch_bf16_cfg<128, 128, 2, 2, 0, Epi, false, true, LEAD>(a, epi, s);
    if (a.N % 256 == 0 && !a.a_map) {   // (the 256x256 fp8 kernel takes no row gather)
        constexpr int NCU = 256;
        const int tn = a.N / 256, t256 = ((a.M + 255) / 256) * tn, rounds = t256 / NCU, rem = t256 - rounds * NCU;
        if (rounds >= 1 || rem >= 3 * NCU / 4) {
            if (rem == 0 || rem >= 3 * NCU / 4) return launch_bf16_cfg<256, 256, 2, 4, 0, Epi, false, true, LEAD>(a, epi, s);
            const int body = (rounds * NCU / tn) * 256;
            int rc = launch_bf16_cfg<256, 256, 2, 4, 0, Epi, false, true, LEAD>(a, epi, s, 0, body);
            if (rc) return rc;
            return launch_bf16_cfg<128, 128, 2, 2, 0, Epi, false, true, LEAD>(a, epi, s, body, a.M);
        }
    }
    return launch_bf16_cfg<128, 128, 2, 2, 0, Epi, false, true, LEAD>(a, epi, s);
}

// wide-N GEMM against a pre-shuffled frozen weight -> gemm_bf16_bpre_kernel (see run_bf16)
static bool takes_bpre(const GemmArgs& a, bool k768 = false) {
    return a.Wp && g_use_bpre && a.N % 256 == 0 && a.K % 256 == 0 && (a.N >= g_big_tile_min_n || k768) && a.M >= 2048;
}
template <class Epi, bool CAT = false, int LEAD = 0>
static int run_bf16(const GemmArgs& a, const Epi& epi, hipStream_t s) {
    if (a.K % 64 != 0 || a.M <= 0) { set_error("gemm_bf16: K=%d must be a multiple of 64, M=%d", a.K, a.M); return -1; }
    if constexpr (SplitKEpi<Epi>::value) {
        // the K = 3072 GEMMs of the cls-only last block: split-K over 256-wide slices + a reduce launch that runs the functor (gemm_skinny.h).
        // B = 128, serial: fc2 forward 56 -> see DESIGN.md 7d; DYT_OPT_GEMM_SPLITK 0 keeps the 6-tile launches
        const int splitk = get_gemm_splitk();
        const int slices = a.K / SK_SLICE + (CAT ? 1 : 0);
        if (splitk && a.splitk_ws && a.M <= SK_MAX_M && a.K >= 1024 && a.K % SK_SLICE == 0 && a.N % 64 == 0 && !a.m_dev && !a.a_fold && !a.a_ld && !a.f8 &&
            (size_t)slices * a.M * a.N * sizeof(float) <= a.splitk_ws_bytes && (!CAT || (a.A2 && a.W2))) {
            hipLaunchKernelGGL((gemm_splitk_kernel<CAT>), dim3(a.N / 32, (a.M + 127) / 128, slices), dim3(256), 0, s, static_cast<const bf16*>(a.A),
                               static_cast<const bf16*>(a.W), a.M, a.N, a.K, a.a_map, static_cast<const bf16*>(a.A2), static_cast<const bf16*>(a.W2),
                               a.a2_map, a.splitk_ws);   // (not counted in g_bf16_kernel_launches: tools/pmc_traffic.py selects the gemm_bf16_* kernels)
            hipLaunchKernelGGL((splitk_reduce_kernel<Epi>), dim3((unsigned)(((size_t)a.M * (a.N / 4) + 255) / 256)), dim3(256), 0, s, a.splitk_ws, slices, a.M,
                               a.N, a.out_scale, epi);
            DYT_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    if constexpr (CAT) {
        if (!a.A2 || !a.W2 || a.N % 128 != 0) { set_error("gemm_bf16: K-concatenated form needs A2, W2 and N %% 128 == 0 (N=%d)", a.N); return -1; }
        // (Round 6: this form through the pre-shuffled-weight kernel -- the leading tile in its prologue's DMA round, bit-identical results -- measured
        // 175 us serial against 123 + 28 us here and 23.4 vs 23.2 ms in the step: the fp32 read-modify-write epilogue of a 128x256 tile does not fit
        // beside 128 accumulators -- 20-37 spilled registers.  Not kept; profiles/round6/r6_fc2_bpre_ab.txt.)
    } else if constexpr (LEAD > 0) {
        if (!a.A2 || !a.W2 || a.N % 128 != 0 || !a.a_fold) { set_error("gemm_bf16: leading three-part tiles need A2, W2, the split form and N %% 128 == 0 (N=%d)", a.N); return -1; }
    } else {
        if (a.A2) { set_error("gemm_bf16: this epilogue has no K-concatenated form"); return -1; }
    // Wide-N GEMMs against a frozen weight: the pre-shuffled-weight kernel (128x256 tiles, two workgroups per CU, the
    // weight never touches LDS).  In the step: 28.10 vs 28.40 ms with the 256x256 kernel; routing the N = 768 GEMMs
    // through it as well gains another 0.5 % wall time but costs 8 % serial GEMM time, so they keep the split-row scheme.
    // K = 768, N = 768 plain-store GEMMs (proj dgrad) take this kernel as well: with only 12 k-steps the 256x256 kernel's exposed
    // prologue / epilogue (one workgroup per CU) weighs most: 45 vs 53.5 us serial.  (DYT_BPRE_K768: 0 off, 1 = every K = 768 GEMM
    // with a pre-shuffled weight, i.e. the proj forward too -- measured 81.7 vs 73.3 us for that one, and the row kernels that read
    // its fp32 output right after it got slower; 2 = default: those without a residual epilogue)
    static const int bpre_k768 = getenv("DYT_BPRE_K768") ? atoi(getenv("DYT_BPRE_K768")) : 2;
    // DYT_BPRE_STORE_MAXK: plain-store N = 768 GEMMs up to this K take the kernel as well.  Round 5: 3072 = the qkv dgrad (K = 2304) and the fc1 dgrad
    // (K = 3072, compacted in the student pass) too.  In the serial profile they are slower there (GEMM family 20.2 -> 20.7 ms per step: the
    // 256x256 kernel has the better main loop on warm operands), but the step -- what `value` measures -- is 23.85 vs 24.3 ms same-box, four
    // alternations: 64 KB workgroups share CUs with the other pass's kernels, and the four-slot ring loses less on operands that come from
    // HBM (tools/gemm_bench.py COLD=1: +22 % vs +42 %).  768 restores the round-4 routing (proj dgrad only).
    static const int bpre_maxk = getenv("DYT_BPRE_STORE_MAXK") ? atoi(getenv("DYT_BPRE_STORE_MAXK")) : 3072;
    const bool k768 = (a.K == D && bpre_k768 == 1) || (bpre_k768 == 2 && std::is_same<Epi, EpiStoreAT<bf16>>::value && a.K <= bpre_maxk);
    if (takes_bpre(a, k768)) {
        GemmArgs b = a; b.W = a.Wp;
        return launch_bf16_bpre<0>(b, epi, s);
    }
    }
    if constexpr (!CAT) {
        // one-part split GEMMs with a wide N (GELU' dgrad of "fp16x3f": 12 k-tiles against an epilogue that reads gelu' and writes dZ):
        // the 256x256 kernel's exposed epilogue outweighs its main loop -> 128x128 tiles, two workgroups per CU (DYT_SPLIT_SHORTK_SMALL=0: off)
        static const int shortk_small = getenv("DYT_SPLIT_SHORTK_SMALL") ? atoi(getenv("DYT_SPLIT_SHORTK_SMALL")) : 1;   // 2: the N = 768 ones (proj dgrad) too
        if (LEAD == 0 && shortk_small && a.a_ld && a.K <= D && (a.N >= g_big_tile_min_n || shortk_small == 2) && a.N % 128 == 0) return launch_bf16_cfg<128, 128, 2, 2, 0>(a, epi, s);
    }
    if (a.N % 256 == 0 && a.N >= g_big_tile_min_n && a.M >= 2048) return launch_bf16_cfg<256, 256, 2, 4, 0, Epi, CAT, false, LEAD>(a, epi, s);
    if constexpr (!CAT) {
        // K <= 768, N = 768 with a residual epilogue (proj forward, patch embedding): 12 k-steps against an epilogue that moves 194 MB -- the
        // launch is bound by its epilogue traffic, and 1182 tiles of 128x128 on 512 slots interleave main loops and epilogues where 255 big
        // tiles run them as two chip-wide phases: 74.6 vs 59 + 20.5 us (256x256 body + 128x128 row tail), step 24.98 vs 25.03 ms same-box,
        // one launch instead of two; same k order, same bits.  DYT_SHORTK_N768_SMALL=0: the split-row scheme for these too
        static const int shortk_n768 = getenv("DYT_SHORTK_N768_SMALL") ? atoi(getenv("DYT_SHORTK_N768_SMALL")) : 1;
        if (LEAD == 0 && shortk_n768 && a.K <= D && a.N % 128 == 0 && a.M >= 2048 && !a.a_ld) return launch_bf16_cfg<128, 128, 2, 2, 0>(a, epi, s);
    }
    if (a.N % 256 == 0 && a.K >= 256 && g_split_rows) {
        // Narrow-N GEMMs (N = 768): per row, 256x256 tiles are ~1.6x cheaper than 128x128 tiles (half the L2->LDS bytes
        // per FLOP), but 99 x 3 = 297 tiles leave 41 for a second round.  The rows that fill whole rounds of 256
        // CUs get 256x256 tiles, the remaining rows 128x128 tiles, as two launches on the same stream.  Both kernels
        // accumulate every dot product in the same k order, so results do not depend on where the split falls.
        // Compacted launches (device-side row count) leave the tail launch empty.  Measured in the step: 28.5 vs
        // 29.1 ms (all-128x128) vs 28.8 ms (all-256x256, two rounds).
        constexpr int NCU = 256;
        const int tn = a.N / 256, t256 = ((a.M + 255) / 256) * tn, rounds = t256 / NCU, rem = t256 - rounds * NCU;
        if (rounds >= 1 || rem >= 3 * NCU / 4) {
            if (rem == 0 || rem >= 3 * NCU / 4) return launch_bf16_cfg<256, 256, 2, 4, 0, Epi, CAT, false, LEAD>(a, epi, s);
            const int body = (rounds * NCU / tn) * 256;
            // DYT_GEMM_ROWS_ONE_LAUNCH=1: both tile shapes in one launch (gemm_bf16_rows_kernel).  Measured: serial step 29.48 -> 28.92 ms
            // (no second launch, no empty tail launches of compacted GEMMs), but the overlapped step 26.2-26.4 -> 26.7 ms: the tail
            // workgroups then hold 128 KB of LDS like the big ones and keep the other pass's 64 KB kernels off their CUs.  Off.
            static const bool one_launch = getenv("DYT_GEMM_ROWS_ONE_LAUNCH") && atoi(getenv("DYT_GEMM_ROWS_ONE_LAUNCH"));
            if constexpr (LEAD == 0) { if (one_launch) return launch_bf16_rows<Epi, CAT>(a, epi, s, body); }
            int rc = launch_bf16_cfg<256, 256, 2, 4, 0, Epi, CAT, false, LEAD>(a, epi, s, 0, body);
            if (rc) return rc;
            return launch_bf16_cfg<128, 128, 2, 2, 0, Epi, CAT, false, LEAD>(a, epi, s, body, a.M);
        }
    }
    if (a.N % 128 == 0) return launch_bf16_cfg<128, 128, 2, 2, 0, Epi, CAT, false, LEAD>(a, epi, s);
    if constexpr (!CAT && LEAD == 0) { if (a.N % 64 == 0) return launch_bf16_cfg<128, 64, 2, 2, 0>(a, epi, s); }
    set_error("gemm_bf16: N=%d must be a multiple of 64", a.N);
    return -1;
}

template <int BM, int BN, class Epi>
static int launch_f32_cfg(const GemmArgs& a, const Epi& epi, hipStream_t s) {
    const int grid = ((a.M + BM - 1) / BM) * (a.N / BN);
    const size_t lds = 2 * (BM + BN) * 128;
    auto kern = gemm_f32_mfma_nt_kernel<BM, BN, 2, 2, Epi>;
    static bool attr_set[64] = {};
    int dev = 0;
    DYT_HIP_CHECK(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        DYT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
        attr_set[dev & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, static_cast<const float*>(a.A), static_cast<const float*>(a.W),
                       a.M, a.N, a.K, a.m_dev, a.a_map, 0, epi);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <class Epi>
static int run_f32(const GemmArgs& a, const Epi& epi, hipStream_t s) {
    if (a.K % 64 != 0 || a.N % 64 != 0 || a.M <= 0) {
        set_error("gemm_f32: N=%d %% 64, K=%d %% 64 required, M=%d", a.N, a.K, a.M);
        return -1;
    }
    if (a.N % 128 == 0) return launch_f32_cfg<128, 128>(a, epi, s);
    return launch_f32_cfg<128, 64>(a, epi, s);
}

// epilogues of the forward GEMMs against frozen weights: the ones the fp8-correction kernels are built for
template <class Epi> struct F8Epi : std::false_type {};
template <> struct F8Epi<EpiQKV<float>> : std::true_type {};
template <> struct F8Epi<EpiQKVPlanes> : std::true_type {};
template <class OT> struct F8Epi<EpiBiasResid<float, OT>> : std::true_type {};
template <bool G, class GT, bool F> struct F8Epi<EpiFc1<float, G, GT, F>> : std::true_type {};
template <bool P, class HT> struct F8Epi<EpiFc2<float, P, HT>> : std::true_type {};
template <> struct F8Epi<EpiAdUp<true>> : std::true_type {};    // cls-row proj of the teacher's last block
template <> struct F8Epi<EpiEmbed> : std::true_type {};
template <> struct F8Epi<EpiBiasF32> : std::true_type {};       // unit entry dyt_linear_split
// SPLIT: fp32 epilogue functors on the 16-bit MFMA kernels (the fp32 operands arrive as K-concatenated 16-bit hi / lo parts)
template <class AT, bool SPLIT, class Epi, int LEAD = 0>
static int run(const GemmArgs& a, const Epi& epi, hipStream_t s) {
    if constexpr (SPLIT && F8Epi<Epi>::value) { if (a.f8) return run_f8<Epi, LEAD>(a, epi, s); }
    if (a.f8) { set_error("gemm: this epilogue has no fp8-correction form"); return -1; }
    if constexpr (sizeof(AT) == 2 || SPLIT) return run_bf16<Epi, false, LEAD>(a, epi, s);
    else return run_f32(a, epi, s);
}

template <class AT, bool SPLIT = false>
static int dispatch(EpiKind kind, const GemmArgs& a, hipStream_t s) {
    switch (kind) {
        case EPI_BIAS_F32: return run<AT, SPLIT>(a, EpiBiasF32{a.bias, a.out_f32, a.N}, s);
        case EPI_QKV:
            if constexpr (SPLIT) {
                if (a.qkv_lo[0]) return run<AT, SPLIT>(a, EpiQKVPlanes{a.bias, (bf16*)a.out_at, (bf16*)a.out_at2, (bf16*)a.out_at3, (bf16*)a.qkv_lo[0],
                                                                     (bf16*)a.qkv_lo[1], (bf16*)a.qkv_lo[2]}, s);
            }
            return run<AT, SPLIT>(a, EpiQKV<AT>{a.bias, (AT*)a.out_at, (AT*)a.out_at2, (AT*)a.out_at3}, s);
        case EPI_BIAS_RESID:
            if constexpr (SPLIT) { if (a.save16) return run<AT, SPLIT>(a, EpiBiasResid<AT, bf16>{a.bias, a.resid, a.out_f32, (bf16*)a.out_at, a.N, nullptr, a.row_scale}, s); }
            if constexpr (sizeof(AT) == 2) {
                if (a.ln_part) {
                    if (a.N != LN_PARTS * 64) { set_error("gemm: LayerNorm partials need N = %d", LN_PARTS * 64); return -1; }
                    return run<AT, SPLIT>(a, EpiBiasResid<AT, AT, true>{a.bias, a.resid, a.out_f32, (AT*)a.out_at, a.N, a.ln_part, a.row_scale}, s);
                }
            }
            if (a.ln_part) { set_error("gemm: LayerNorm partials exist in the 16-bit modes only"); return -1; }
            return run<AT, SPLIT>(a, EpiBiasResid<AT>{a.bias, a.resid, a.out_f32, (AT*)a.out_at, a.N, nullptr, a.row_scale}, s);
        case EPI_FC1:
            if constexpr (SPLIT) {   // the split forms take the A&S GELU (EpiFc1<FAST>): 42.7 vs 43.1 ms/step, logits vs the oracle unchanged (2.4e-5 / 6e-6)
                if (a.save16 && a.out_at2) return run<AT, SPLIT>(a, EpiFc1<AT, true, bf16, true>{a.bias, (AT*)a.out_at, (bf16*)a.out_at2, a.N, (bf16*)a.out3, a.out3_f8}, s);
                if (a.out_at2) return run<AT, SPLIT>(a, EpiFc1<AT, true, AT, true>{a.bias, (AT*)a.out_at, (AT*)a.out_at2, a.N, (bf16*)a.out3, a.out3_f8}, s);
                return run<AT, SPLIT>(a, EpiFc1<AT, false, AT, true>{a.bias, (AT*)a.out_at, nullptr, a.N, (bf16*)a.out3, a.out3_f8}, s);
            }
            if constexpr (sizeof(AT) == 2) {
                if (a.ln_part) {   // LayerNorm folded in
                    if (!a.ln_cs || !a.ln_st_out || !a.ln_scratch || a.K != LN_PARTS * 64) { set_error("gemm: folded LayerNorm form needs ln_cs, ln_st_out, ln_scratch and K = 768"); return -1; }
                    const float2* st = nullptr;
                    if (!takes_bpre(a)) {   // no statistics prologue in these tile shapes: merge the partials in a pre-pass
                        hipLaunchKernelGGL(ln_finalize_kernel, dim3((a.M + 255) / 256), dim3(256), 0, s, a.ln_part, a.a_map, a.a_map ? a.ln_scratch : a.ln_st_out,
                                           a.a_map ? a.ln_st_out : nullptr, a.M);
                        DYT_HIP_CHECK(hipGetLastError());
                        st = a.a_map ? a.ln_scratch : a.ln_st_out;
                    }
                    if (a.out_at2) return run<AT, SPLIT>(a, EpiFc1<AT, true, AT, false, true>{a.bias, (AT*)a.out_at, (AT*)a.out_at2, a.N, nullptr, 0, a.ln_cs, st, a.ln_part, a.ln_st_out}, s);
                    return run<AT, SPLIT>(a, EpiFc1<AT, false, AT, false, true>{a.bias, (AT*)a.out_at, nullptr, a.N, nullptr, 0, a.ln_cs, st, a.ln_part, a.ln_st_out}, s);
                }
            }
            if (a.ln_part) { set_error("gemm: the folded LayerNorm form exists in the 16-bit modes only"); return -1; }
            if (a.out_at2) return run<AT, SPLIT>(a, EpiFc1<AT, true>{a.bias, (AT*)a.out_at, (AT*)a.out_at2, a.N, (bf16*)a.out3, a.out3_f8}, s);
            return run<AT, SPLIT>(a, EpiFc1<AT, false>{a.bias, (AT*)a.out_at, nullptr, a.N, (bf16*)a.out3, a.out3_f8}, s);
        case EPI_FC2: {
            const float* resid = a.resid ? a.resid : a.out_f32;   // null: in place
            if (a.A2) {   // adapter up-projection as the leading k-tile of the contraction (16-bit kernels only)
                if constexpr (sizeof(AT) == 2) {
                    if (!a.row_map && !a.row_mask)
                        return run_bf16<EpiFc2<AT, true>, true>(a, EpiFc2<AT, true>{a.bias, a.out_f32, nullptr, nullptr, (AT*)a.h_out, resid, a.bias2, a.scale, a.row_scale}, s);
                    return run_bf16<EpiFc2<AT, false>, true>(a, EpiFc2<AT, false>{a.bias, a.out_f32, a.row_map, a.row_mask, (AT*)a.h_out, resid, a.bias2, a.scale, a.row_scale}, s);
                } else if constexpr (SPLIT) {
                    // split fp32 forms whose backward runs on 16-bit operands: the up-projection as three leading tiles of a three-part product (CatArgs: LEAD)
                    if (!a.save16) {   // passes that save nothing (evaluation): the MLP output, if wanted at all, in fp32
                        if (!a.row_map && !a.row_mask)
                            return run<AT, SPLIT, EpiFc2<AT, true>, 3>(a, EpiFc2<AT, true>{a.bias, a.out_f32, nullptr, nullptr, (AT*)a.h_out, resid, a.bias2, a.scale, a.row_scale}, s);
                        return run<AT, SPLIT, EpiFc2<AT, false>, 3>(a, EpiFc2<AT, false>{a.bias, a.out_f32, a.row_map, a.row_mask, (AT*)a.h_out, resid, a.bias2, a.scale, a.row_scale}, s);
                    }
                    if (!a.row_map && !a.row_mask)
                        return run<AT, SPLIT, EpiFc2<AT, true, bf16>, 3>(a, EpiFc2<AT, true, bf16>{a.bias, a.out_f32, nullptr, nullptr, (bf16*)a.h_out, resid, a.bias2, a.scale, a.row_scale}, s);
                    return run<AT, SPLIT, EpiFc2<AT, false, bf16>, 3>(a, EpiFc2<AT, false, bf16>{a.bias, a.out_f32, a.row_map, a.row_mask, (bf16*)a.h_out, resid, a.bias2, a.scale, a.row_scale}, s);
                } else {
                    set_error("gemm: the K-concatenated fc2 form exists in the 16-bit and the 16-bit-backward split modes only");
                    return -1;
                }
            }
            if constexpr (SPLIT) {
                if (a.save16 && a.h_out) {
                    if (!a.row_map && !a.row_mask) return run<AT, SPLIT>(a, EpiFc2<AT, true, bf16>{a.bias, a.out_f32, nullptr, nullptr, (bf16*)a.h_out, resid, nullptr, 0.f, a.row_scale}, s);
                    return run<AT, SPLIT>(a, EpiFc2<AT, false, bf16>{a.bias, a.out_f32, a.row_map, a.row_mask, (bf16*)a.h_out, resid, nullptr, 0.f, a.row_scale}, s);
                }
            }
            if (!a.row_map && !a.row_mask) return run<AT, SPLIT>(a, EpiFc2<AT, true>{a.bias, a.out_f32, nullptr, nullptr, (AT*)a.h_out, resid, nullptr, 0.f, a.row_scale}, s);
            return run<AT, SPLIT>(a, EpiFc2<AT, false>{a.bias, a.out_f32, a.row_map, a.row_mask, (AT*)a.h_out, resid, nullptr, 0.f, a.row_scale}, s);
        }
        case EPI_GELU_BWD:
            if (a.row_map) return run<AT, SPLIT>(a, EpiGeluBwd<AT, true>{(const AT*)a.aux_at, (AT*)a.out_at, a.N, a.row_map, (bf16*)a.out3, a.out3_scale, a.out3_hi_only}, s);
            return run<AT, SPLIT>(a, EpiGeluBwd<AT, false>{(const AT*)a.aux_at, (AT*)a.out_at, a.N, nullptr, (bf16*)a.out3, a.out3_scale, a.out3_hi_only}, s);
        case EPI_STORE_F32: return run<AT, SPLIT>(a, EpiStoreF32{a.out_f32, a.N, a.accumulate, a.scale}, s);
        case EPI_STORE_AT: return run<AT, SPLIT>(a, EpiStoreAT<AT>{(AT*)a.out_at, a.N}, s);
        case EPI_AD_DOWN:
            if constexpr (!SPLIT && sizeof(AT) == 4) {
                if (a.save16) return run<AT, SPLIT>(a, EpiAdDown<AT, bf16>{a.bias, (AT*)a.out_at, a.keep, a.r, a.inv_keep, a.drop_p, a.seed, a.subseq, a.row_map, a.seed_dev, (bf16*)a.out_at2, a.scale, (bf16*)a.out3, a.out3_scale}, s);
            }
            return run<AT, SPLIT>(a, EpiAdDown<AT>{a.bias, (AT*)a.out_at, a.keep, a.r, a.inv_keep, a.drop_p, a.seed, a.subseq, a.row_map, a.seed_dev, (AT*)a.out_at2, a.scale, (bf16*)a.out3, a.out3_scale}, s);
        case EPI_AD_UP:
            if (a.row_map) return run<AT, SPLIT>(a, EpiAdUp<true>{a.bias, a.resid, a.out_f32, a.scale, a.row_map, nullptr, a.row_scale, a.bias_scale}, s);
            return run<AT, SPLIT>(a, EpiAdUp<false>{a.bias, a.resid, a.out_f32, a.scale, nullptr, a.row_mask, nullptr, a.bias_scale}, s);
        case EPI_AD_DGRAD_UP:
            return run<AT, SPLIT>(a, EpiAdDgradUp<AT>{(const AT*)a.aux_at, (AT*)a.out_at, a.scale, a.inv_keep}, s);
        case EPI_EMBED: return run<AT, SPLIT>(a, EpiEmbed{a.bias, a.pos, a.out_f32}, s);
        case EPI_BIAS_AT: return run<AT, SPLIT>(a, EpiBiasAT<AT>{a.bias, (AT*)a.out_at, a.N}, s);
    }
    set_error("gemm: unknown epilogue %d", (int)kind);
    return -1;
}

// measurement hook: plain bf16 GEMM into a bf16 C with a selectable kernel variant
int launch_gemm_raw(const void* A, const void* W, void* C, int M, int N, int K, int variant, hipStream_t s) {
    if (K % 64 != 0 || N % 256 != 0 || M <= 0) { set_error("gemm_raw: bad shape"); return -1; }
    GemmArgs a; a.A = A; a.W = W; a.M = M; a.N = N; a.K = K;
    EpiStoreAT<bf16> epi{static_cast<bf16*>(C), N};
    switch (variant) {
        case 0: return launch_bf16_cfg<128, 128, 2, 2, 0>(a, epi, s);
        case 9: return launch_bf16_cfg<128, 128, 2, 2, 9>(a, epi, s);     // + phase timers
        case 10: return launch_bf16_cfg<256, 256, 2, 4, 0>(a, epi, s);
        case 19: return launch_bf16_cfg<256, 256, 2, 4, 9>(a, epi, s);
        case 30: return run_bf16(a, epi, s);                               // the product dispatch (incl. the split-row scheme)
        // residual epilogues with the phase timers: C is read as the fp32 residual stream [M,N] (the same bytes as bf16 [2M,N]), updated in place
        case 21: return launch_bf16_cfg<256, 256, 2, 4, 9>(a, EpiFc2<bf16, true>{nullptr, static_cast<float*>(C), nullptr, nullptr, nullptr, static_cast<const float*>(C), nullptr, 0.f}, s);
        case 22: return launch_bf16_cfg<128, 128, 2, 2, 9>(a, EpiBiasResid<bf16>{nullptr, static_cast<const float*>(C), static_cast<float*>(C), nullptr, N}, s);
        case 23: return launch_bf16_cfg<256, 256, 2, 4, 9>(a, EpiBiasResid<bf16>{nullptr, static_cast<const float*>(C), static_cast<float*>(C), nullptr, N}, s);
        case 40: case 41: case 42: {   // C = A2 W2^T + A W^T, A2 [M,64] stored behind A, W2 [N,64] behind W (leading k-tile form)
            a.A2 = static_cast<const bf16*>(A) + (size_t)M * K; a.W2 = static_cast<const bf16*>(W) + (size_t)N * K;
            if (variant == 40) return launch_bf16_cfg<128, 128, 2, 2, 0, EpiStoreAT<bf16>, true>(a, epi, s);
            if (variant == 41) return launch_bf16_cfg<256, 256, 2, 4, 0, EpiStoreAT<bf16>, true>(a, epi, s);
            return run_bf16<EpiStoreAT<bf16>, true>(a, epi, s);
        }
        case 70: case 79: {   // pre-shuffled-weight kernel (test-only: shuffles W into a cached scratch buffer first)
            static bf16* wp = nullptr; static size_t wp_elems = 0;
            const size_t need = (size_t)N * K;
            if (need > wp_elems) {
                if (wp) (void)hipFree(wp);
                DYT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&wp), need * 2));
                wp_elems = need;
            }
            if (launch_preshuffle_w(W, wp, N, K, s)) return -2;
            GemmArgs b = a; b.W = wp;
            if (variant == 70) return launch_bf16_bpre<0>(b, epi, s);
            return launch_bf16_bpre<9>(b, epi, s);
        }
    }
    set_error("gemm_raw: unknown variant %d", variant);
    return -1;
}

// measurement hook: fp32 GEMM (exact-fp32 MFMA kernel, product dispatch) into an fp32 C [M,N]
// variant 0: plain store; 1: EpiAdUp reading the residual from C + M*N and writing C; 2: EpiStoreF32 accumulate (C += A W^T)
int launch_gemm_f32_raw(const void* A, const void* W, void* C, int M, int N, int K, int variant, hipStream_t s) {
    GemmArgs a; a.A = A; a.W = W; a.M = M; a.N = N; a.K = K;
    float* c = static_cast<float*>(C);
    if (variant == 1) return run_f32(a, EpiAdUp<false>{nullptr, c + (size_t)M * N, c, 0.1f, nullptr, nullptr}, s);
    if (variant == 2) return run_f32(a, EpiStoreF32{c, N, 1, 1.0f}, s);
    return run_f32(a, EpiStoreAT<float>{c, N}, s);
}

int gemm_debug_counters(unsigned long long* out4, int reset) {
    DYT_HIP_CHECK(hipDeviceSynchronize());
    DYT_HIP_CHECK(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_gemm_dbg), 4 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[4] = {0, 0, 0, 0};
        DYT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg), z, sizeof(z)));
    }
    return 0;
}

// ---- fp32 operands as 16-bit hi / lo parts ----
// A [M,K] fp32 (rows optionally gathered) -> out [M, 3K] = [hi | hi | lo]; one thread = 8 consecutive k of a row
__global__ __launch_bounds__(256) void split3_a_kernel(const float* __restrict__ A, const int* __restrict__ a_map,
                                                       const int* __restrict__ m_dev, bf16* __restrict__ out, int M, int K, float scale, int f8) {
    const int kc = K / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(idx / kc), c = (int)(idx - (size_t)row * kc);
    const int Mv = m_dev ? min(*m_dev, M) : M;
    if (row >= Mv) return;
    const float* src = A + (size_t)(a_map ? a_map[row] : row) * K + c * 8;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src) * scale, x1 = *reinterpret_cast<const f32x4*>(src + 4) * scale;
    if (f8) {   // hi16 / fp8 form (store4_split_f8)
        bf16* rowp = out + (size_t)row * SPLIT_A * K;
        store4_split_f8(rowp, K, c * 8, x0[0], x0[1], x0[2], x0[3]);
        store4_split_f8(rowp, K, c * 8 + 4, x1[0], x1[1], x1[2], x1[3]);
        return;
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const Split2 s0 = split2(x0[i]), s1 = split2(x1[i]);
        hi[i] = s0.hi; lo[i] = s0.lo;
        hi[4 + i] = s1.hi; lo[4 + i] = s1.lo;
    }
    bf16* dst = out + (size_t)row * SPLIT_A * K + c * 8;
    *reinterpret_cast<bf16x8*>(dst) = hi;
    *reinterpret_cast<bf16x8*>(dst + K) = lo;
}
// W [N,K] fp32 -> [N, 2K] = [hi | lo]; the contraction reads it as [hi | lo | hi] (the B loader folds the last third onto the first)
__global__ __launch_bounds__(256) void split3_w_kernel(const float* __restrict__ W, bf16* __restrict__ out, int N, int K) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)N * K) return;
    const size_t row = idx / K; const int k = (int)(idx - row * K);
    const float w = W[idx];
    const Split2 sw = split2(w);
    const bf16 hi = sw.hi, lo = sw.lo;
    bf16* dst = out + row * SPLIT_A * K;
    dst[k] = hi; dst[K + k] = lo;
}
// hi16 / fp8 form of a weight: [N][hi16 (K) | e4m3(lo 2^(ew+11)) (K bytes) | e4m3(hi 2^ew) (K bytes)], ew = 7 - ceil(log2 max|w|)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ W, size_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(W[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bit patterns
}
__device__ __forceinline__ int weight_exp(unsigned maxbits) {
    const float m = __uint_as_float(maxbits);
    if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
    int e;
    const float f = frexpf(m, &e);          // m = f 2^e, f in [0.5, 1): ceil(log2 m) = e, or e - 1 when m is a power of two
    const int cl = f == 0.5f ? e - 1 : e;
    return max(-60, min(60, 7 - cl));
}
__global__ __launch_bounds__(256) void split_w_f8_kernel(const float* __restrict__ W, bf16* __restrict__ out, int N, int K,
                                                         const unsigned* __restrict__ maxbits, int* __restrict__ ew_out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // 4 consecutive k of a row
    const int ew = weight_exp(*maxbits);
    if (idx == 0) *ew_out = ew;
    if (idx >= (size_t)N * K / 4) return;
    const size_t row = idx / (K / 4); const int k = (int)(idx - row * (K / 4)) * 4;
    f32x4 w = *reinterpret_cast<const f32x4*>(W + row * K + k);
    asm("" : "+v"(w));
    const bf16 h0 = (bf16)w[0], h1 = (bf16)w[1], h2 = (bf16)w[2], h3 = (bf16)w[3];
    bf16* rowp = out + row * SPLIT_A * K;
    *reinterpret_cast<bf16x4*>(rowp + k) = bf16x4{h0, h1, h2, h3};
    const float sh = ldexpf(1.0f, ew), sl = ldexpf(1.0f, ew + 11);
    unsigned char* r8 = reinterpret_cast<unsigned char*>(rowp) + 2 * (size_t)K + k;
    *reinterpret_cast<int*>(r8) = pack4_e4m3((w[0] - (float)h0) * sl, (w[1] - (float)h1) * sl, (w[2] - (float)h2) * sl, (w[3] - (float)h3) * sl);
    *reinterpret_cast<int*>(r8 + K) = pack4_e4m3((float)h0 * sh, (float)h1 * sh, (float)h2 * sh, (float)h3 * sh);
}
int launch_split_w_f8(const float* W, void* W3, int N, int K, int* ew_dev, unsigned* scratch, hipStream_t s) {
    DYT_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(unsigned), s));
    const size_t n = (size_t)N * K;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>(1024, (n + 255) / 256)), dim3(256), 0, s, W, n, scratch);
    hipLaunchKernelGGL(split_w_f8_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, W, static_cast<bf16*>(W3), N, K, scratch, ew_dev);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_split3_w(const float* W, void* W3, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(split3_w_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, s, W, static_cast<bf16*>(W3), N, K);
    DYT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemm(int precision, EpiKind kind, const GemmArgs& a, hipStream_t s) {
    if (precision == 0 && a.W3 && a.a3) {
        if (a.K % 64 != 0 || (a.A2 && kind != EPI_FC2)) { set_error("gemm split form: K=%d %% 64", a.K); return -1; }
        const size_t tasks = (size_t)a.M * (a.K / 8);
        if (!a.a3_ready)
        hipLaunchKernelGGL(split3_a_kernel, dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, s, static_cast<const float*>(a.A), a.a_map,
                           a.m_dev, static_cast<bf16*>(a.a3), a.M, a.K, a.a3_scale, a.f8 ? 1 : 0);
        GemmArgs b = a;
        // a3_parts products of the contraction: 3 = hi*hi + hi*lo + lo*hi, 2 = hi * (hi + lo) (A rounded to half, W exact), 1 = hi*hi
        b.A = a.a3; b.W = a.W3; b.K = a.a3_parts * a.K; b.a_fold = a.K / 64; b.a_ld = 2 * a.K; b.a_map = (a.a3_ready && a.a3_mapped) ? a.a_map : nullptr; b.Wp = nullptr; b.W3 = nullptr; b.out_scale = 1.0f / a.a3_scale;
        if (a.f8) {   // hi * hi as f16 tiles, then the two correction products as fp8 tiles along the same rows
            if (a.K % 128 != 0 || a.a3_parts != 3 || a.a3_scale != 1.0f || !a.w_exp) { set_error("gemm f8 form: K=%d %% 128, forward operands only", a.K); return -1; }
            b.K = 2 * a.K; b.a_fold = 0; b.f8_begin = a.K / 64;
        }
        return dispatch<float, true>(kind, b, s);
    }
    if (dbg_skip(64) && (a.K == RP || a.N == RP)) return 0;
    if (dbg_skip(128) && a.N == D && a.K >= 256) return 0;
    if (dbg_skip(256) && a.N >= 2304) return 0;
    return precision == 0 ? dispatch<float>(kind, a, s) : dispatch<bf16>(kind, a, s);
}

}  // namespace dyt
