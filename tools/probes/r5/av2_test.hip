// Stand-alone check of attention_v2.hip's kernels against a double-precision host reference (1 image = 12 heads).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDYT_FP16=1 -I dynamic-tuning_amd/csrc tools/probes/r5/av2_test.hip -o tools/probes/r5/av2_test
#include "attention_v2.hip"
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include <random>
namespace dyt { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
using namespace dyt;
__global__ void fill_kernel(bf16* p, size_t n, unsigned seed, float scale) {   // distinct pseudo-random values (sum of 4 uniforms ~ normal)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ (seed * 0x9E3779B9u); float a = 0.f;
        for (int j = 0; j < 4; ++j) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; a += (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        p[i] = (bf16)(a * 1.73f * scale);
    }
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0 random, 1 q = 0 (uniform softmax), 2 v[key][d] = key, 3 v[key][d] = d
    const int B = 1, nh = B * NH;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<bf16> q(nh * NT * HD), k(nh * NT * HD), v(nh * NT * HD);
    std::vector<double> qd(q.size()), kd(q.size()), vd(q.size());
    for (size_t i = 0; i < q.size(); ++i) {
        const int d = i % HD, key = (i / HD) % NT;
        float a = nd(rng) * 0.4f, b = nd(rng) * 1.5f, c = nd(rng);
        if (mode == 1) a = 0.f;
        if (mode == 2) c = (float)key;
        if (mode == 3) c = (float)d;
        q[i] = (bf16)a; k[i] = (bf16)b; v[i] = (bf16)c;
        qd[i] = (double)(float)q[i]; kd[i] = (double)(float)k[i]; vd[i] = (double)(float)v[i];
    }
    bf16 *dq, *dk, *dv, *dout; float* dlse;
    hipMalloc(&dq, q.size() * 2); hipMalloc(&dk, q.size() * 2); hipMalloc(&dv, q.size() * 2);
    hipMalloc(&dout, (size_t)B * NT * D * 2); hipMalloc(&dlse, nh * NT * 4);
    hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dk, k.data(), q.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dv, v.data(), q.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, (size_t)B * NT * D * 2);
    int rc = launch_attn_fwd_v2(dq, dk, dv, dout, dlse, B, 0);
    hipError_t e = hipDeviceSynchronize();
    printf("launch rc %d sync %s\n", rc, hipGetErrorString(e));
    {   // determinism: repeat the forward launch and compare bitwise
        std::vector<bf16> o0((size_t)B * NT * D), o1((size_t)B * NT * D);
        hipMemcpy(o0.data(), dout, o0.size() * 2, hipMemcpyDeviceToHost);
        for (int rep = 0; rep < 8; ++rep) {
            hipMemset(dout, 0xff, (size_t)B * NT * D * 2);
            if (rep < 4) launch_attn_fwd_v2(dq, dk, dv, dout, dlse, B, 0);
            else hipLaunchKernelGGL((av2::attn_fwd_v2_kernel<false, 0>), dim3(nh), dim3(512), 5 * av2::IMG, 0, dq, dk, dv, dout, dlse, nh);
            hipDeviceSynchronize();
            hipMemcpy(o1.data(), dout, o1.size() * 2, hipMemcpyDeviceToHost);
            int nd = 0;
            for (size_t i = 0; i < o0.size(); ++i)
                if (__builtin_bit_cast(unsigned short, o0[i]) != __builtin_bit_cast(unsigned short, o1[i])) {
                    if (nd < 6) printf("  fwd rep %d differs at row %zu head %zu d %zu: %g vs %g\n", rep, i / D, (i % D) / HD, i % HD, (double)(float)o0[i], (double)(float)o1[i]);
                    ++nd;
                }
            printf("fwd rep %d: %d of %zu elements differ\n", rep, nd, o0.size());
            if (rep < 4) for (int hh = 0; hh < NH; ++hh) {
                int cnt[7] = {0}; bool any = false;
                for (int r = 0; r < NT; ++r) for (int d = 0; d < HD; ++d) { size_t i = (size_t)r * D + hh * HD + d;
                    if (__builtin_bit_cast(unsigned short, o0[i]) != __builtin_bit_cast(unsigned short, o1[i])) { cnt[r / 32]++; any = true; } }
                if (any) printf("     head %d: per 32-row tile %d %d %d %d %d %d %d\n", hh, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6]);
            }
        }
    }
    std::vector<bf16> out((size_t)B * NT * D); std::vector<float> lse(nh * NT);
    hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(lse.data(), dlse, lse.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0, maxlse = 0; int bad = 0;
    for (int bh = 0; bh < nh; ++bh) {
        const int b = bh / NH, h = bh % NH;
        for (int i = 0; i < NT; ++i) {
            std::vector<double> s(NT); double mx = -1e300;
            for (int j = 0; j < NT; ++j) { double a = 0; for (int d = 0; d < HD; ++d) a += qd[((size_t)bh * NT + i) * HD + d] * kd[((size_t)bh * NT + j) * HD + d]; s[j] = a; mx = std::max(mx, a); }
            double sum = 0; for (int j = 0; j < NT; ++j) { s[j] = std::exp(s[j] - mx); sum += s[j]; }
            maxlse = std::max(maxlse, std::fabs((mx + std::log(sum)) - (double)lse[(size_t)bh * NT + i]));
            for (int d = 0; d < HD; ++d) {
                double o = 0; for (int j = 0; j < NT; ++j) o += s[j] * vd[((size_t)bh * NT + j) * HD + d];
                o /= sum;
                const double g = (double)(float)out[((size_t)b * NT + i) * D + h * HD + d];
                const double err = std::fabs(g - o);
                maxref = std::max(maxref, std::fabs(o));
                if (!(err <= 0.02 * std::max(1.0, std::fabs(o)))) { if (bad < 24) printf("bh %d q %d d %d got %g ref %g\n", bh, i, d, g, o); ++bad; }
                if (err == err) maxerr = std::max(maxerr, err);
            }
        }
    }
    if (argc > 2) {   // timing at B = atoi(argv[2]): full kernel and the ablations
        const int Bb = atoi(argv[2]), nhb = Bb * NH;
        bf16 *bq, *bk, *bv, *bo; float* bl;
        const size_t n = (size_t)nhb * NT * HD;
        hipMalloc(&bq, n * 2); hipMalloc(&bk, n * 2); hipMalloc(&bv, n * 2); hipMalloc(&bo, n * 2); hipMalloc(&bl, (size_t)nhb * NT * 4);
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, bq, n, 1u, 0.4f);
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, bk, n, 2u, 1.5f);
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, bv, n, 3u, 1.0f);
        char* flush; hipMalloc(&flush, 512u << 20);
        const size_t lds = 5 * av2::IMG;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char* name, auto kern) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            float best = 1e9, tot = 0;
            for (int it = 0; it < 12; ++it) {
                if (argc > 3 && argv[3][0] == 'c') hipMemsetAsync(flush, it, 512u << 20, 0);   // cold caches
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kern, dim3(std::min(nhb, 256)), dim3(512), lds, 0, bq, bk, bv, bo, bl, nhb);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) { best = std::min(best, ms); tot += ms; }
            }
            printf("%-28s best %.1f us  avg %.1f us\n", name, best * 1e3, tot * 100);
        };
        if (argc > 5) {   // backward timing instead
            bf16 *bdo, *bg;
            hipMalloc(&bdo, n * 2); hipMalloc(&bg, n * 6);
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, bdo, n, 4u, 1.0f);
            hipLaunchKernelGGL((av2::attn_fwd_v2_kernel<true, 0>), dim3(std::min(nhb, 256)), dim3(512), lds, 0, bq, bk, bv, bo, bl, nhb);
            auto runb = [&](const char* name, auto kern) {
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)av2::BWD_LDS);
                float best = 1e9, tot = 0;
                for (int it = 0; it < 12; ++it) {
                    if (argc > 3 && argv[3][0] == 'c') hipMemsetAsync(flush, it, 512u << 20, 0);
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(kern, dim3(std::min(nhb, 256)), dim3(512), av2::BWD_LDS, 0, bq, bk, bv, bo, bdo, bl, bg, nhb, 7, D);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (it >= 2) { best = std::min(best, ms); tot += ms; }
                }
                printf("bwd %-28s best %.1f us  avg %.1f us\n", name, best * 1e3, tot * 100);
            };
            {   // determinism at steady state: four launches, bitwise
                std::vector<unsigned short> g0(n * 3), g1(n * 3);
                hipFuncSetAttribute((const void*)av2::attn_bwd_v2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)av2::BWD_LDS);
                for (int rep = 0; rep < 4; ++rep) {
                    hipMemset(bg, 0xff, n * 6);
                    hipLaunchKernelGGL((av2::attn_bwd_v2_kernel<0>), dim3(std::min(nhb, 256)), dim3(512), av2::BWD_LDS, 0, bq, bk, bv, bo, bdo, bl, bg, nhb, 7, D);
                    hipDeviceSynchronize();
                    hipMemcpy(rep ? g1.data() : g0.data(), bg, n * 6, hipMemcpyDeviceToHost);
                    if (rep) { size_t nd = 0; for (size_t i = 0; i < g0.size(); ++i) nd += g0[i] != g1[i]; printf("bwd B=%d rep %d: %zu of %zu elements differ\n", Bb, rep, nd, g0.size()); }
                }
            }
            auto stamps = [&](const char* name, auto kern) {
                unsigned long long z[24] = {0}, r[24];
                hipMemcpyToSymbol(HIP_SYMBOL(av2::g_bwd_dbg), z, sizeof(z));
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)av2::BWD_LDS);
                for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(std::min(nhb, 256)), dim3(512), av2::BWD_LDS, 0, bq, bk, bv, bo, bdo, bl, bg, nhb, 7, D);
                hipDeviceSynchronize();
                hipMemcpyFromSymbol(r, HIP_SYMBOL(av2::g_bwd_dbg), sizeof(r));
                const double nhd = (double)r[5];
                printf("stamps %-20s per head (100 MHz ticks?): wait(a) %.0f  phase B %.0f  wait(c) %.0f  phase A %.0f  dq stores %.0f  (heads %.0f)\n", name, r[0] / nhd, r[1] / nhd, r[2] / nhd, r[3] / nhd, r[4] / nhd, nhd);
                printf("   per wave phase B:"); for (int w = 0; w < 7; ++w) printf(" %.0f", r[8 + w] / nhd); printf("   phase A:"); for (int w = 0; w < 7; ++w) printf(" %.0f", r[16 + w] / nhd); printf("\n");
            };
            stamps("full", av2::attn_bwd_v2_kernel<8>);
            stamps("no stats", av2::attn_bwd_v2_kernel<8 + 16>);
            stamps("no stats no rows", av2::attn_bwd_v2_kernel<8 + 16 + 32>);
            stamps("no stats no rows no drain", av2::attn_bwd_v2_kernel<8 + 16 + 32 + 64>);
            stamps("compute only", av2::attn_bwd_v2_kernel<13>);
            runb("full", av2::attn_bwd_v2_kernel<0>);
            runb("full, full-drain loader", av2::attn_bwd_v2_kernel<0, 0, 0>);
            {   // determinism of the LT form + equality with the default form
                std::vector<unsigned short> g0(n * 3), g1(n * 3);
                auto kd = av2::attn_bwd_v2_kernel<0, 0, 0>; auto kl = av2::attn_bwd_v2_kernel<0, 0, 1>;
                hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)av2::BWD_LDS);
                hipLaunchKernelGGL(kd, dim3(std::min(nhb, 256)), dim3(512), av2::BWD_LDS, 0, bq, bk, bv, bo, bdo, bl, bg, nhb, 7, D);
                hipDeviceSynchronize(); hipMemcpy(g0.data(), bg, n * 6, hipMemcpyDeviceToHost);
                for (int rep = 0; rep < 6; ++rep) {
                    hipMemset(bg, 0xff, n * 6);
                    hipLaunchKernelGGL(kl, dim3(std::min(nhb, 256)), dim3(512), av2::BWD_LDS, 0, bq, bk, bv, bo, bdo, bl, bg, nhb, 7, D);
                    hipDeviceSynchronize(); hipMemcpy(g1.data(), bg, n * 6, hipMemcpyDeviceToHost);
                    size_t nd = 0; for (size_t i = 0; i < g0.size(); ++i) nd += g0[i] != g1[i];
                    printf("bwd LT rep %d vs default form: %zu elements differ\n", rep, nd);
                }
            }
            runb("full prio alt", av2::attn_bwd_v2_kernel<0, 2>);
            runb("full prio mfma", av2::attn_bwd_v2_kernel<0, 3>);
            runb("no DMA, no stores, prio alt", av2::attn_bwd_v2_kernel<5, 2>);
            runb("no stores", av2::attn_bwd_v2_kernel<1>);
            runb("no compute", av2::attn_bwd_v2_kernel<2>);
            runb("no compute, no stores", av2::attn_bwd_v2_kernel<3>);
            runb("no DMA", av2::attn_bwd_v2_kernel<4>);
            runb("no DMA, no stores", av2::attn_bwd_v2_kernel<5>);
            runb("nothing", av2::attn_bwd_v2_kernel<7>);
            return 0;
        }
        run("full (pipe)", av2::attn_fwd_v2_kernel<true, 0>);
        run("full (no pipe)", av2::attn_fwd_v2_kernel<false, 0>);
        run("no stores", av2::attn_fwd_v2_kernel<true, 1>);
        run("no compute", av2::attn_fwd_v2_kernel<true, 2>);
        run("no compute, no stores", av2::attn_fwd_v2_kernel<true, 3>);
        run("no DMA", av2::attn_fwd_v2_kernel<true, 4>);
        run("no DMA, no stores", av2::attn_fwd_v2_kernel<true, 5>);
        run("nothing (q loads+barriers)", av2::attn_fwd_v2_kernel<true, 7>);
    }
    {   // backward (B = 1): dq, dk, dv vs double
        std::vector<bf16> dO((size_t)B * NT * D);
        for (auto& x : dO) x = (bf16)(nd(rng));
        bf16 *ddo, *dg;
        hipMalloc(&ddo, dO.size() * 2); hipMalloc(&dg, (size_t)B * NT * 3 * D * 2);
        hipMemcpy(ddo, dO.data(), dO.size() * 2, hipMemcpyHostToDevice);
        hipMemset(dg, 0xff, (size_t)B * NT * 3 * D * 2);
        const int nqt = argc > 4 ? atoi(argv[4]) : 7;
        int rcb = launch_attn_bwd_v2(dq, dk, dv, dout, ddo, dlse, dg, B, 0, nqt, D);
        hipError_t eb = hipDeviceSynchronize();
        printf("bwd launch rc %d sync %s\n", rcb, hipGetErrorString(eb));
        std::vector<bf16> g((size_t)B * NT * 3 * D);
        hipMemcpy(g.data(), dg, g.size() * 2, hipMemcpyDeviceToHost);
        double me[3] = {0, 0, 0}, mr[3] = {0, 0, 0}; int badb = 0;
        for (int bh = 0; bh < nh; ++bh) {
            const int b = bh / NH, h = bh % NH;
            std::vector<double> P((size_t)NT * NT), dS((size_t)NT * NT);
            for (int i = 0; i < NT; ++i) {
                double mx = -1e300, sum = 0;
                for (int j = 0; j < NT; ++j) { double a = 0; for (int d = 0; d < HD; ++d) a += qd[((size_t)bh * NT + i) * HD + d] * kd[((size_t)bh * NT + j) * HD + d]; P[(size_t)i * NT + j] = a; mx = std::max(mx, a); }
                for (int j = 0; j < NT; ++j) { P[(size_t)i * NT + j] = std::exp(P[(size_t)i * NT + j] - mx); sum += P[(size_t)i * NT + j]; }
                double delta = 0;
                std::vector<double> dP(NT);
                for (int j = 0; j < NT; ++j) {
                    P[(size_t)i * NT + j] /= sum;
                    double a = 0;
                    for (int d = 0; d < HD; ++d) a += (nqt == 7 || i < 32 * nqt ? (double)(float)dO[((size_t)b * NT + i) * D + h * HD + d] : 0.0) * vd[((size_t)bh * NT + j) * HD + d];
                    dP[j] = a; delta += P[(size_t)i * NT + j] * a;
                }
                for (int j = 0; j < NT; ++j) dS[(size_t)i * NT + j] = P[(size_t)i * NT + j] * (dP[j] - delta);
            }
            for (int i = 0; i < NT; ++i)
                for (int d = 0; d < HD; ++d) {
                    double rq = 0, rk = 0, rv = 0;
                    for (int j = 0; j < NT; ++j) {
                        rq += dS[(size_t)i * NT + j] * kd[((size_t)bh * NT + j) * HD + d];
                        rk += dS[(size_t)j * NT + i] * qd[((size_t)bh * NT + j) * HD + d];
                        rv += P[(size_t)j * NT + i] * (nqt == 7 || j < 32 * nqt ? (double)(float)dO[((size_t)b * NT + j) * D + h * HD + d] : 0.0);
                    }
                    const double ref[3] = {rq * 0.125, rk, rv};
                    for (int t = 0; t < 3; ++t) {
                        const double got = (double)(float)g[((size_t)b * NT + i) * 3 * D + t * D + h * HD + d];
                        const double err = std::fabs(got - ref[t]);
                        mr[t] = std::max(mr[t], std::fabs(ref[t]));
                        if (!(err <= 0.03 * std::max(0.3, std::fabs(ref[t])))) { if (badb < 16) printf("bh %d %s row %d d %d got %g ref %g\n", bh, t == 0 ? "dq" : t == 1 ? "dk" : "dv", i, d, got, ref[t]); ++badb; }
                        if (err == err) me[t] = std::max(me[t], err);
                    }
                }
        }
        { unsigned long long hsh = 1469598103934665603ull; for (auto x : g) { hsh = (hsh ^ (unsigned short)__builtin_bit_cast(unsigned short, x)) * 1099511628211ull; }
          unsigned long long h2 = 1469598103934665603ull; for (auto x : out) { h2 = (h2 ^ (unsigned short)__builtin_bit_cast(unsigned short, x)) * 1099511628211ull; }
          printf("hash fwd out %016llx  bwd dqkv %016llx\n", h2, hsh); }
        printf("bwd (nq %d): dq err %.3e / %.3f  dk err %.3e / %.3f  dv err %.3e / %.3f  bad %d\n", nqt, me[0], mr[0], me[1], mr[1], me[2], mr[2], badb);
    }
    printf("mode %d: max abs err %.3e (max |ref| %.3f), lse err %.3e, bad %d\n", mode, maxerr, maxref, maxlse, bad);
    return 0;
}
