#!/usr/bin/env python3
"""Micro-benchmark of the exact-fp32 MFMA GEMM (csrc/gemm_f32_mfma.h) on the path's shapes (MI355X); every row is
checked against an fp64 product of the same fp32 operands."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
from _lib import check, lib, ptr, stream_ptr  # noqa: E402

SHAPES = [(25216, 2304, 768), (25216, 768, 768), (25216, 3072, 768), (25216, 768, 3072), (17690, 3072, 768),
          (17690, 768, 3072), (25216, 768, 2304), (25216, 64, 768), (25216, 768, 64)]

def epilogue_variants(iters):
    """the adapter-sized K = 64 GEMM with its three epilogues (time is epilogue / memory, not MFMA)"""
    M, N, K = 25216, 768, 64
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    c = torch.zeros(2 * M, N, device="cuda")
    for variant, name in ((0, "plain store"), (1, "adapter-up (resid + 0.1 acc)"), (2, "accumulate")):
        for _ in range(2):
            check(lib().dyt_gemm_f32_raw(ptr(a), ptr(w), ptr(c), M, N, K, variant, stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            check(lib().dyt_gemm_f32_raw(ptr(a), ptr(w), ptr(c), M, N, K, variant, stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        print("%d,%d,%d %-30s %8.1f us" % (M, N, K, name, e0.elapsed_time(e1) / iters * 1e3), flush=True)


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    epilogue_variants(iters)
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        c = torch.empty(M, N, device="cuda")
        for _ in range(2):
            check(lib().dyt_gemm_f32_raw(ptr(a), ptr(w), ptr(c), M, N, K, 0, stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            check(lib().dyt_gemm_f32_raw(ptr(a), ptr(w), ptr(c), M, N, K, 0, stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        rows = torch.arange(0, M, 97, device="cuda")
        ref = (a[rows].double() @ w.double().t())
        err = float((c[rows].double() - ref).abs().max() / ref.abs().max())
        full = float((c - a @ w.t()).abs().max() / ref.abs().max())   # every row vs torch's fp32 GEMM
        print("%-20s %8.1f us  %6.1f TFLOP/s  (%.3f of 157.3)  err vs fp64 %.1e  all rows vs torch fp32 %.1e" % (
            "%d,%d,%d" % (M, N, K), ms * 1e3, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 157.3, err, full), flush=True)
