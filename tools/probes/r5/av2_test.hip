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
                if (argc > 3) hipMemsetAsync(flush, it, 512u << 20, 0);   // cold caches
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kern, dim3(std::min(nhb, 256)), dim3(512), lds, 0, bq, bk, bv, bo, bl, nhb);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) { best = std::min(best, ms); tot += ms; }
            }
            printf("%-28s best %.1f us  avg %.1f us\n", name, best * 1e3, tot * 100);
        };
        run("full (pipe)", av2::attn_fwd_v2_kernel<true, 0>);
        run("full (no pipe)", av2::attn_fwd_v2_kernel<false, 0>);
        run("no stores", av2::attn_fwd_v2_kernel<true, 1>);
        run("no compute", av2::attn_fwd_v2_kernel<true, 2>);
        run("no compute, no stores", av2::attn_fwd_v2_kernel<true, 3>);
        run("no DMA", av2::attn_fwd_v2_kernel<true, 4>);
        run("no DMA, no stores", av2::attn_fwd_v2_kernel<true, 5>);
        run("nothing (q loads+barriers)", av2::attn_fwd_v2_kernel<true, 7>);
    }
    printf("mode %d: max abs err %.3e (max |ref| %.3f), lse err %.3e, bad %d\n", mode, maxerr, maxref, maxlse, bad);
    return 0;
}
