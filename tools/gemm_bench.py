#!/usr/bin/env python3
"""Micro-benchmark of the bf16 GEMM kernel variants on the path's shapes (MI355X)."""
import os
import sys

import torch

DBG = (9, 19, 79, 21, 22, 23)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
from _lib import check, lib, ptr, stream_ptr  # noqa: E402

SHAPES = [(25216, 2304, 768), (25216, 768, 768), (25216, 3072, 768), (25216, 768, 3072), (17690, 3072, 768),
          (17690, 768, 3072), (25216, 768, 2304), (21760, 768, 3072)]


def bench(M, N, K, variant, iters=20):
    a = (torch.randn(M, K, device="cuda") * 1.0).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    c = torch.empty(2 * M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        check(lib().dyt_gemm_bf16_raw(ptr(a), ptr(w), ptr(c), M, N, K, variant, stream_ptr()))
    torch.cuda.synchronize()
    if variant in DBG:
        import ctypes
        check(lib().dyt_debug_counters((ctypes.c_uint64 * 4)(), 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get("COLD"):   # every launch on operands that are NOT in the 256 MB Infinity Cache: a 1 GiB fill between launches, per-launch events
        flush = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
        tot = 0.0
        for _ in range(iters):
            flush.fill_(1.0)
            e0.record()
            check(lib().dyt_gemm_bf16_raw(ptr(a), ptr(w), ptr(c), M, N, K, variant, stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        ms = tot / iters
        del flush
    else:
        e0.record()
        for _ in range(iters):
            check(lib().dyt_gemm_bf16_raw(ptr(a), ptr(w), ptr(c), M, N, K, variant, stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    if variant in DBG:
        import ctypes
        buf = (ctypes.c_uint64 * 4)()
        check(lib().dyt_debug_counters(buf, 1))
        n = max(1, buf[3])
        print("      v%d phases (cycles/WG): prologue %.0f  main %.0f  epilogue %.0f  (WGs %d)" % (variant, buf[0] / n, buf[1] / n, buf[2] / n, n))
    err = None
    if variant in CHECKED:
        ref = (a.float() @ w.float().t())   # every row: races only show at full occupancy
        err = float((c[:M].float() - ref).abs().max() / ref.abs().max())
    return ms, err


if __name__ == "__main__":
    variants = [int(v) for v in sys.argv[1:]] or [0, 10, 70, 30]
    CHECKED = {v for v in variants if v in (0, 10, 30, 70)}
    print("%-22s" % "M,N,K" + "".join("  v%-2d us / TF/s (err)     " % v for v in variants))
    for (M, N, K) in SHAPES:
        line = "%-22s" % ("%d,%d,%d" % (M, N, K))
        for v in variants:
            ms, err = bench(M, N, K, v)
            line += "  %7.1f / %6.1f %-9s" % (ms * 1e3, 2.0 * M * N * K / ms / 1e9, "" if err is None else "(%.1e)" % err)
        print(line, flush=True)
