"""Self-concurrency check of the bf16 GEMM kernels: the same launch on two streams at once, repeated; every result must equal
the serial one bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch
from _lib import check, lib, ptr
L = lib()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (M, N, K, v) in [(788, 768, 768, 0), (788, 768, 3072, 0), (788, 3072, 768, 0), (788, 64, 768, 0), (788, 768, 64, 0), (788, 768, 2304, 0),
                     (3152, 768, 768, 30), (3152, 2304, 768, 70), (3152, 768, 3072, 30)]:
    torch.manual_seed(M + N + K)
    a = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(2)]
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    ref = []
    for i in range(2):
        c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        check(L.dyt_gemm_bf16_raw(ptr(a[i]), ptr(w), ptr(c), M, N, K, v, 0))
        torch.cuda.synchronize()
        ref.append(c.clone())
    bad = 0
    outs = [[torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(50)] for _ in range(2)]
    for rep in range(4):
        for j in range(50):
            check(L.dyt_gemm_bf16_raw(ptr(a[0]), ptr(w), ptr(outs[0][j]), M, N, K, v, s1.cuda_stream))
            check(L.dyt_gemm_bf16_raw(ptr(a[1]), ptr(w), ptr(outs[1][j]), M, N, K, v, s2.cuda_stream))
        torch.cuda.synchronize()
        for i in range(2):
            for j in range(50):
                if not torch.equal(outs[i][j], ref[i]):
                    bad += 1
    print("M=%d N=%d K=%d v=%d: %d of 400 concurrent results differ from the serial one" % (M, N, K, v, bad), flush=True)
