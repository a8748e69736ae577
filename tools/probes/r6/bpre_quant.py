#!/usr/bin/env python3
"""Tile-count quantisation of gemm_bf16_bpre_kernel (128x256 tiles, two workgroups per CU = 512 slots): time of the N = 768 dgrad shapes
against the number of row tiles.  variant 70 re-shuffles W before every launch (a 6 us kernel): timed separately and subtracted."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
from _lib import check, lib, ptr, stream_ptr

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

N = 768
for K in (768, 2304, 3072):
    torch.manual_seed(0)
    Mmax = 8 * 21760
    a = torch.randn(Mmax, K, device="cuda").half().view(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).half().view(torch.bfloat16)
    c = torch.zeros(2 * Mmax, N, device="cuda", dtype=torch.bfloat16)
    # the shuffle alone: a launch with M = 1 tile
    base = timeit(lambda: check(lib().dyt_gemm_bf16_raw(ptr(a), ptr(w), ptr(c), 128, N, K, 70, stream_ptr())))
    for rows_t in (85, 128, 138, 170, 171, 197, 256, 341, 342, 512, 1360):
        M = rows_t * 128
        t = timeit(lambda: check(lib().dyt_gemm_bf16_raw(ptr(a), ptr(w), ptr(c), M, N, K, 70, stream_ptr())))
        tiles = rows_t * 3
        print("K=%4d row tiles %4d tiles %4d (%.2f rounds of 512): %7.1f us (1-tile launch %.1f)  %.3f PFLOP/s  us per 512 tiles %.1f" % (
            K, rows_t, tiles, tiles / 512, t, base, 2 * M * N * K / (t - base + 3) / 1e9, (t - base + 3) / tiles * 512))
