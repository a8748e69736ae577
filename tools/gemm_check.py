#!/usr/bin/env python3
"""Bitwise determinism / cross-variant equality of the bf16 GEMM variants (same k-order accumulation)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
from _lib import check, lib, ptr, stream_ptr
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 10

def run(a, w, M, N, K, v):
    c = torch.zeros(2 * M, N, device="cuda", dtype=torch.bfloat16)
    check(lib().dyt_gemm_bf16_raw(ptr(a), ptr(w), ptr(c), M, N, K, v, stream_ptr()))
    torch.cuda.synchronize()
    return c[:M].clone()

for (M, N, K) in [(25216, 2304, 768), (25216, 3072, 768), (17690, 3072, 768), (4000, 2304, 768), (25216, 768, 3072)]:
    torch.manual_seed(0)
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    ref = run(a, w, M, N, K, 0)
    bad_rows_total = 0
    for rep in range(6):
        c = run(a, w, M, N, K, VAR)
        diff = (c != ref)
        nbad = int(diff.sum())
        if nbad:
            rows = diff.any(dim=1).nonzero()[:, 0]
            cols = diff.any(dim=0).nonzero()[:, 0]
            print("  rep %d: %d mismatching elements, rows %d..%d (%d rows), cols %d..%d, max abs %.4f" % (
                rep, nbad, int(rows.min()), int(rows.max()), rows.numel(), int(cols.min()), int(cols.max()),
                float((c.float() - ref.float()).abs().max())))
            bad_rows_total += nbad
    print("M=%d N=%d K=%d: v%d vs v0 bitwise %s" % (M, N, K, VAR, "EQUAL (6 reps)" if bad_rows_total == 0 else "DIFFERENT"))
