"""GPU-measured data point for the 'split 16-bit x 3' question: one GEMM of the path computed as
   hi*hi + hi*lo + lo*hi with 16-bit MFMA operands (A = A_hi + A_lo, W = W_hi + W_lo; the three products as ONE contraction over the
   K-concatenated operands [A_hi | A_hi | A_lo] x [W_hi | W_lo | W_hi]^T through the product's own kernels), next to the plain 16-bit
   GEMM and the exact-fp32 MFMA GEMM: error vs fp64 (dyt_linear, fp32 output) and time (raw hooks, the bench shapes).
   usage: python tools/probes/split_precision_probe.py   (both operand types)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch
import _lib
from _lib import check, ptr, stream_ptr


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


for fp16 in (True, False):
    L = _lib.lib(fp16=fp16)
    dt = torch.float16 if fp16 else torch.bfloat16
    name = "fp16" if fp16 else "bf16"
    for (M, N, K, what) in ((25216, 2304, 768, "qkv"), (25216, 768, 3072, "fc2")):
        g = torch.Generator(device="cuda").manual_seed(M + N)
        a = torch.randn(M, K, device="cuda", generator=g) * float(os.environ.get("PA_SCALE", "1"))   # PA_SCALE=0.05: lo parts in the fp16 subnormals
        w = torch.randn(N, K, device="cuda", generator=g) * 0.03
        a_hi, w_hi = a.to(dt).float(), w.to(dt).float()
        a_lo, w_lo = (a - a_hi).to(dt).float(), (w - w_hi).to(dt).float()
        rows = 2048
        ref = (a[:rows].double() @ w.double().t())
        scale = float(ref.abs().max())

        def lin(x, ww, prec):
            c = torch.empty(x.shape[0], ww.shape[0], device="cuda")
            check(L.dyt_linear(ptr(x.contiguous()), ptr(ww.contiguous()), None, ptr(c), x.shape[0], ww.shape[0], x.shape[1], prec, stream_ptr()))
            torch.cuda.synchronize()
            return c
        e_plain = float((lin(a[:rows], w, 1).double() - ref).abs().max()) / scale
        acat = torch.cat([a_hi, a_hi, a_lo], dim=1)[:rows]
        wcat = torch.cat([w_hi, w_lo, w_hi], dim=1)
        e_split = float((lin(acat, wcat, 1).double() - ref).abs().max()) / scale
        e_f32 = float((lin(a[:rows], w, 0).double() - ref).abs().max()) / scale
        # time: raw hooks at the full shape (16-bit C for the 16-bit kernels, fp32 C for the fp32 kernel)
        A16, W16 = a.to(dt).contiguous(), w.to(dt).contiguous()
        A3 = torch.cat([a_hi, a_hi, a_lo], dim=1).to(dt).contiguous()
        W3 = wcat.to(dt).contiguous()
        C16 = torch.empty(M, N, device="cuda", dtype=dt)
        C32 = torch.empty(M, N, device="cuda")
        t_plain = timeit(lambda: check(L.dyt_gemm_bf16_raw(ptr(A16), ptr(W16), ptr(C16), M, N, K, 30, stream_ptr())))
        t_split = timeit(lambda: check(L.dyt_gemm_bf16_raw(ptr(A3), ptr(W3), ptr(C16), M, N, 3 * K, 30, stream_ptr())))
        t_f32 = timeit(lambda: check(L.dyt_gemm_f32_raw(ptr(a), ptr(w), ptr(C32), M, N, K, 0, stream_ptr())))
        fl = 2.0 * M * N * K
        print("%s %-3s [%d x %d x %d]  max err / max|C|: plain %.2e  split-x3 %.2e  fp32-MFMA %.2e   time: plain %.0f us (%.0f TF/s)  split-x3 %.0f us (%.1fx, %.0f TF/s useful)  fp32-MFMA %.0f us (%.0f TF/s)"
              % (name, what, M, N, K, e_plain, e_split, e_f32, t_plain, fl / t_plain / 1e6, t_split, t_split / t_plain, fl / t_split / 1e6, t_f32, fl / t_f32 / 1e6))
