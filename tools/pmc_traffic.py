#!/usr/bin/env python3
"""HBM traffic per launch of a kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB-units of the L2's memory-side
requests; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read
(128-B requests tallied at 64 B) -> it is DOUBLED here; WRITE_SIZE is taken as reported (uncalibrated).
Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [name-substring]
"""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        s = acc.setdefault(k, [0, 0.0])
        s[0] += 1
        s[1] += float(r["Counter_Value"])
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
sub = sys.argv[3] if len(sys.argv) > 3 else "gemm_bf16_nt_kernel"
n = sum(v[0] for k, v in fetch.items() if sub in k)
fk = sum(v[1] for k, v in fetch.items() if sub in k)
wk = sum(v[1] for k, v in write.items() if sub in k)
out = {"kernel_family": sub, "launches": n,
       "fetch_bytes_per_launch_corrected": 2.0 * fk * 1024 / max(n, 1),
       "write_bytes_per_launch": wk * 1024 / max(n, 1),
       "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024 / max(n, 1),
       "correction": "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KiB -> bytes"}
print(json.dumps(out))
tot_f = sum(v[1] for v in fetch.values()); tot_w = sum(v[1] for v in write.values())
print("all kernels: fetch(corrected) %.2f GB, write %.2f GB over the profiled run" % (2 * tot_f * 1024 / 1e9, tot_w * 1024 / 1e9), file=sys.stderr)
