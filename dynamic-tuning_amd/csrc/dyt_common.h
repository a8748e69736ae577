// Internal shared definitions for libdyt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// cache hints for single-use streams; -DDYT_NO_NT builds the library without them (A/B and determinism probes)
#ifdef DYT_NO_NT
#define DYT_NT_LOAD(p) (*(p))
#define DYT_NT_STORE(v, p) (*(p) = (v))
#else
#define DYT_NT_LOAD(p) __builtin_nontemporal_load(p)
#define DYT_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

// Full drain of the wave's vector-memory counter, ORDERED AFTER the loads that produced the listed registers.
//
// Why: on gfx950 a partial wait (`s_waitcnt vmcnt(N)`, N > 0 -- what the compiler emits when a value is needed while younger
// loads are still in flight) is only safe if loads complete in issue order.  In the row kernels of the backward pass that
// assumption does not hold when the CU is shared with another stream's LDS/MFMA-heavy workgroups: a load that hits (LN / gate
// weights, resident in L1) completes ahead of an older load that misses (the row's statistics, its residual snapshot), the
// counter reaches N, and the wave consumes the older load's destination register before the data has landed -- i.e. whatever
// a previous wave left there, typically another row's value of the same quantity.  Results then differ run to run at the
// 1e-2 level in single rows (found with CU-masked streams, per-class isolation, NaN poisoning and
// `-mllvm -amdgpu-waitcnt-forcezero`; tools/probes/determinism_*.py).  A drain that is (i) a full `vmcnt(0)` and (ii) carries
// one register of every load instruction of the group as an operand -- so that neither the scheduler nor a `__restrict__`
// qualifier can move a load below it -- removes the exposure at no measurable cost (the kernels are HBM-bound).
#define DYT_VMEM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DYT_PIN1(a) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a) : : "memory")
#define DYT_PIN2(a, b) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory")
#define DYT_PIN3(a, b, c) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) : : "memory")
#define DYT_PIN4(a, b, c, d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory")

namespace dyt {

constexpr int D = 768;        // embed dim
constexpr int NH = 12;        // heads
constexpr int HD = 64;        // head dim
constexpr int NT = 197;       // tokens per image (cls + 196 patches)
constexpr int NP = 196;       // patch tokens
constexpr int DM = 3072;      // MLP hidden
constexpr int RP = 64;        // adapter bottleneck padded to one MFMA N-tile
constexpr float LN_EPS = 1e-6f;

// The 16-bit operand type of the "fast" kernels.  The library is built twice from the same sources:
//   libdyt_hip.so      bfloat16 operands (8 exponent / 7 mantissa bits)          -- precision "bf16"
//   libdyt_hip_f16.so  IEEE half operands (-DDYT_FP16: 5 / 10 bits; the reference's own GPU dtype, it trains under fp16 autocast,
//                      engine_finetune.py:47) with the gradient stream scaled by 2^k wherever it is held in 16 bits -- precision "fp16"
// `bf16` / `bf16x8` below are the names of that operand type in BOTH builds (kernel names keep "bf16" too).
#ifdef DYT_FP16
typedef _Float16 bf16;
#define DYT_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define DYT_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#else
typedef __bf16 bf16;
#define DYT_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define DYT_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <class T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// store 4 consecutive values
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
    bf16x4 v = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
    *reinterpret_cast<bf16x4*>(p) = v;
}
struct Split2 { bf16 hi, lo; };
// The source value is pinned in ONE register first.  Without that the compiler folds the 16-bit conversion into the operation that
// produced x in some uses (v_fma_mix*: a single rounding of the exact product) and not in others (v_cvt_pk_f16_f32 of the fp32 result:
// two roundings), so the hi part that is stored / fed to the MFMA and the hi part lo is taken against can sit one 16-bit ulp apart
// whenever the fp32 rounding crosses a 16-bit rounding boundary: x != hi + lo by a whole ulp of hi for about one value in 10^4.
__device__ __forceinline__ Split2 split2(float x) {
    asm("" : "+v"(x));
    Split2 r;
    r.hi = (bf16)x;
    r.lo = (bf16)(x - (float)r.hi);
    return r;
}
// The split A operand of a GEMM in memory: [row][hi (K) | lo (K)].  The contraction runs over 3K -- [A_hi | A_hi | A_lo] against the
// weight image [W_hi | W_lo | W_hi] -- and the kernel reads the hi part for the first TWO thirds (GemmArgs / CatArgs::a_fold), so the
// producers write, and HBM holds, two 16-bit copies per value instead of three.
constexpr int SPLIT_A = 2;
// 4 consecutive fp32 values as the hi / lo parts of that layout; dst = &row[col], the lo part K elements further
__device__ __forceinline__ void store4_split3(bf16* dst, int K, float a, float b, float c, float d) {
    const Split2 sa = split2(a), sb = split2(b), sc = split2(c), sd = split2(d);
    const bf16x4 hi = {sa.hi, sb.hi, sc.hi, sd.hi};
    const bf16x4 lo = {sa.lo, sb.lo, sc.lo, sd.lo};
    *reinterpret_cast<bf16x4*>(dst) = hi;
    *reinterpret_cast<bf16x4*>(dst + K) = lo;
}
// ... with the lo half left unwritten when its consumer contracts over the hi part alone (gradient operands of the "fp16x3f" form)
__device__ __forceinline__ void store4_split3(bf16* dst, int K, float a, float b, float c, float d, bool hi_only) {
    if (!hi_only) { store4_split3(dst, K, a, b, c, d); return; }
    const bf16x4 hi = {split2(a).hi, split2(b).hi, split2(c).hi, split2(d).hi};
    *reinterpret_cast<bf16x4*>(dst) = hi;
}
// ---- "fp16f8" form of a split operand: hi * hi on the f16 matrix cores, the two correction products hi * lo + lo * hi as fp8 (e4m3)
// MFMAs (v_mfma_scale_f32_16x16x128_f8f6f4: twice the f16 rate per k) -- they only need ~4 significant bits to keep the result at 2^-15.
// Row image, 4K bytes like the [hi | lo] form:   A: [hi16 (K halfs) | e4m3(hi) (K bytes) | e4m3(lo * 2^12) (K bytes)]
//                                                W: [hi16 (K halfs) | e4m3(lo * 2^(ew+11)) | e4m3(hi * 2^ew)]    ew: per-matrix exponent
// so that the contraction simply continues along the row: k-tiles [0, K/64) are f16 tiles of 64, the next K/64 tiles are fp8 tiles of
// 128 (A_hi8 x W_lo8, then A_lo8 x W_hi8), descaled by the MFMA's E8M0 scale operand (2^-(ew+11), 2^-(ew+12)).
// lo = x - hi is taken in fp32 (exact); activation magnitudes up to 224 keep a full-precision lo part, beyond that it saturates (the
// element then has plain fp16 accuracy); v_cvt_pk_fp8_f32 rounds to nearest even and returns NaN from 480 up, hence the clamp.
constexpr float F8_LO_SCALE = 4096.0f;   // 2^12 on the activations' lo part
constexpr int F8_LO_LOG2 = 12;
__device__ __forceinline__ int pack4_e4m3(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return v;
}
// 4 consecutive fp32 values of an A operand row in that form; row = row base of the image, col = first of the 4 columns
__device__ __forceinline__ void store4_split_f8(bf16* row, int K, int col, float a, float b, float c, float d) {
    asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));   // see split2: one rounding of ONE value
    const bf16 ha = (bf16)a, hb = (bf16)b, hc = (bf16)c, hd = (bf16)d;
    const bf16x4 hi = {ha, hb, hc, hd};
    *reinterpret_cast<bf16x4*>(row + col) = hi;
    unsigned char* r8 = reinterpret_cast<unsigned char*>(row) + 2 * (size_t)K + col;
    *reinterpret_cast<int*>(r8) = pack4_e4m3((float)ha, (float)hb, (float)hc, (float)hd);
    *reinterpret_cast<int*>(r8 + K) = pack4_e4m3((a - (float)ha) * F8_LO_SCALE, (b - (float)hb) * F8_LO_SCALE, (c - (float)hc) * F8_LO_SCALE,
                                                 (d - (float)hd) * F8_LO_SCALE);
}
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void load4(const bf16* p, float (&o)[4]) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}

// Wave-wide reductions on the VALU only: DPP lane permutes inside a 16-lane row, v_readlane across the four rows
// (__shfl_xor lowers to ds_bpermute_b32, five dependent trips through the LDS queue per reduction; the row kernels do two
// to four reductions per 768-channel row).  Every lane receives the total.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_f32(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
#ifdef DYT_DPP_NOPS
// A/B build (tools/probes): every cross-lane step as ONE inline-asm block padded with s_nop 4 on both sides, so that neither the
// scheduler nor the hazard recogniser decides the distance between a VALU write and the DPP / v_readlane read of that register
#define DYT_DPP_ADD(out, in, ctrl)                                                                                   \
    asm volatile("s_nop 4\n\tv_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(out) : "v"(in))
__device__ __forceinline__ float wave_sum(float v) {
    float a, b, c, d;
    DYT_DPP_ADD(a, v, "quad_perm:[1,0,3,2]");
    DYT_DPP_ADD(b, a, "quad_perm:[2,3,0,1]");
    DYT_DPP_ADD(c, b, "row_half_mirror");
    DYT_DPP_ADD(d, c, "row_mirror");
    int s0, s1, s2, s3;
    asm volatile("s_nop 4\n\tv_readlane_b32 %0, %4, 0\n\tv_readlane_b32 %1, %4, 16\n\tv_readlane_b32 %2, %4, 32\n\tv_readlane_b32 %3, %4, 48\n\ts_nop 4"
                 : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(d));
    return (__builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1)) + (__builtin_bit_cast(float, s2) + __builtin_bit_cast(float, s3));
}
#else
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_f32<0x141>(v);   // row_half_mirror: quad <-> neighbouring quad
    v += dpp_f32<0x140>(v);   // row_mirror: half row <-> half row  => every lane holds its row's sum
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
#endif
// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15): every lane of the row receives it
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}
// LayerNorm folded into the GEMM behind it (GemmArgs::ln_part): the producer epilogue leaves LN_PARTS (sum, M2 about the group's own mean)
// pairs per row, one per 64 columns; merged here in a fixed order (Chan et al.: M2 = sum M2_i + 64 sum (mean_i - mean)^2) -> (mean, rstd)
constexpr int LN_PARTS = 12;
__device__ __forceinline__ float2 ln_merge_parts(const float2* __restrict__ part, int row) {
    const float4* p = reinterpret_cast<const float4*>(part + (size_t)row * LN_PARTS);
    float4 v[LN_PARTS / 2];
#pragma unroll
    for (int i = 0; i < LN_PARTS / 2; ++i) v[i] = p[i];
    float sum = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_PARTS / 2; ++i) { sum += v[i].x; sum += v[i].z; m2 += v[i].y; m2 += v[i].w; }
    const float mean = sum * (1.0f / 768.0f);
#pragma unroll
    for (int i = 0; i < LN_PARTS / 2; ++i) {
        const float d0 = v[i].x * (1.0f / 64.0f) - mean, d1 = v[i].z * (1.0f / 64.0f) - mean;
        m2 = fmaf(64.0f * d0, d0, m2);
        m2 = fmaf(64.0f * d1, d1, m2);
    }
    return make_float2(mean, 1.0f / sqrtf(m2 * (1.0f / 768.0f) + LN_EPS));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}



// Normal CDF by Abramowitz-Stegun 26.2.19: Phi(a) = 1 - (1/2)(1 + d1 a + ... + d6 a^6)^-16 for a >= 0,
// |err| < 1.5e-7 (9e-7 in fp32 arithmetic, checked against scipy.erf on [-8,8]); for x < 0 the tail
// (1/2) p^-16 is used directly (no cancellation).  One v_rcp, no v_exp: ~2/3 the issue cost of the
// erfc-polynomial form, and the GELU math is what bounds the fc1 / GELU' epilogues (tools/gemm_bench.py 20-22).
__device__ __forceinline__ float normal_cdf_fast(float x) {
    const float a = fabsf(x);
    float p = fmaf(0.0000053830f, a, 0.0000488906f);
    p = fmaf(p, a, 0.0000380036f);
    p = fmaf(p, a, 0.0032776263f);
    p = fmaf(p, a, 0.0211410061f);
    p = fmaf(p, a, 0.0498673470f);
    p = fmaf(p, a, 1.0f);
    float q = __builtin_amdgcn_rcpf(p);   // v_rcp_f32 (1 ulp); __frcp_rn expands to a full IEEE division
    q *= q; q *= q; q *= q; q *= q;      // p^-16
    const float half = 0.5f * q;
    return x >= 0.f ? 1.0f - half : half;
}
template <class AT> __device__ __forceinline__ float gelu_fwd(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_fwd<bf16>(float x) { return x * normal_cdf_fast(x); }
template <class AT> __device__ __forceinline__ float gelu_bwd(float x) { return gelu_erf_grad(x); }
template <> __device__ __forceinline__ float gelu_bwd<bf16>(float x) {
    return fmaf(x * 0.39894228040143268f, __expf(-0.5f * x * x), normal_cdf_fast(x));
}
// GELU and its derivative in one go (they share Phi(x)): h = x Phi(x), gp = Phi(x) + x pdf(x).
// The fc1 epilogue stores gp (instead of the pre-activation) for the backward pass, whose epilogue is then
// a plain multiply -- the transcendental work is done once, where Phi is already being computed.
template <class AT> __device__ __forceinline__ void gelu_both(float x, float& h, float& gp) {
    const float phi = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    h = x * phi;
    gp = phi + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
template <> __device__ __forceinline__ void gelu_both<bf16>(float x, float& h, float& gp) {
    // Both outputs need pdf(x); Abramowitz-Stegun 26.2.17 builds Phi from that same pdf:
    //   Phi(a) = 1 - pdf(a) t (b1 + b2 t + ... + b5 t^4),  t = 1 / (1 + 0.2316419 a),  a >= 0   (|err| < 7.5e-8)
    // -> one v_exp, one v_rcp, 9 fma/mul; checked against scipy.erf in fp32: |Phi err| 2.8e-7, |h err| 4.2e-7,
    // |gelu' err| 2.9e-7 on [-10, 10] (5 VALU ops fewer per element than 26.2.19 + a separate pdf; the fc1
    // epilogue is bounded by this arithmetic).
    const float a = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.2316419f, a, 1.0f));
    const float pdf = __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752f, -1.3257480647f));   // exp(-x^2/2) / sqrt(2 pi)
    float p = fmaf(1.330274429f, t, -1.821255978f);
    p = fmaf(p, t, 1.781477937f);
    p = fmaf(p, t, -0.356563782f);
    p = fmaf(p, t, 0.319381530f);
    const float tail = pdf * (t * p);
    const float phi = x >= 0.f ? 1.0f - tail : tail;
    h = x * phi;
    gp = fmaf(x, pdf, phi);
}
// the same for two elements at once, written on 2-vectors so that the polynomial, the products and the final
// h / gelu' run as packed-fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32: two elements per issue slot); only the two
// v_rcp / v_exp and the sign select stay scalar.  Bit-identical to gelu_both<bf16> per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_both_x2(f32x2 x, f32x2& h, f32x2& gp) {
    const f32x2 a = __builtin_elementwise_abs(x);
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.2316419f, 0.2316419f}, a, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 e = __builtin_elementwise_fma(x * x, f32x2{-0.72134752f, -0.72134752f}, f32x2{-1.3257480647f, -1.3257480647f});
    const f32x2 pdf = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    f32x2 p = __builtin_elementwise_fma(f32x2{1.330274429f, 1.330274429f}, t, f32x2{-1.821255978f, -1.821255978f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.781477937f, 1.781477937f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.356563782f, -0.356563782f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.319381530f, 0.319381530f});
    const f32x2 tail = pdf * (t * p);
    const f32x2 om = f32x2{1.0f, 1.0f} - tail;
    const f32x2 phi = {x[0] >= 0.f ? om[0] : tail[0], x[1] >= 0.f ? om[1] : tail[1]};
    h = x * phi;
    gp = __builtin_elementwise_fma(x, pdf, phi);
}
// sigmoid exactly as 1/(1+exp(-x)) in fp32 (the form the reference's y_soft > 0.5 test sees)
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- Philox4x32-10 counter RNG (on-device Gumbel/logistic noise and dropout) ----
struct Philox {
    uint32_t c[4];
    __device__ __forceinline__ Philox(uint64_t seed, uint64_t subseq, uint64_t offset) {
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        c[0] = (uint32_t)offset; c[1] = (uint32_t)(offset >> 32);
        c[2] = (uint32_t)subseq; c[3] = (uint32_t)(subseq >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
            uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
            uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
            c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    // uniform in (0,1)
    __device__ __forceinline__ float u01(int i) const { return ((c[i] >> 8) + 0.5f) * (1.0f / 16777216.0f); }
};

// error plumbing (host)
void set_error(const char* fmt, ...);
#define DYT_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            dyt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                   \
        }                                                                                \
    } while (0)

}  // namespace dyt
