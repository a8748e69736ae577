"""Headroom probe: the step's GEMM shapes through torch.matmul (hipBLASLt / rocBLAS behind it), plain store, fp16 and bf16, to compare with
this library's own kernels of the same shapes (profiles/round5/*shape_times*).  Measurement only; nothing in the product calls torch.matmul."""
import torch, time
dev = torch.device("cuda", 0)
M = 128 * 197
shapes = [("qkv fwd", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072), ("qkv dgrad", M, 768, 2304),
          ("student fc1", 17690, 3072, 768), ("student fc2", 17690, 768, 3072), ("cls fc2", 128, 768, 3072)]
for dt in (torch.float16, torch.bfloat16):
    for name, m, n, k in shapes:
        a = torch.randn(m, k, device=dev, dtype=dt) * 0.1
        w = torch.randn(n, k, device=dev, dtype=dt) * 0.1   # NT: C = A W^T
        wt = w.t().contiguous()                             # NN: C = A Wt
        for form, f in (("NT", lambda: a @ w.t()), ("NN", lambda: a @ wt)):
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for rep in range(3):
                e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            print("%-5s %-12s %s M=%5d N=%4d K=%4d  %7.1f us  %6.0f TFLOP/s" % (str(dt)[6:], name, form, m, n, k, best, 2.0 * m * n * k / best / 1e6), flush=True)
