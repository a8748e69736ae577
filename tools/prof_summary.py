#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats CSV."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"]
    m = re.search(r"gemm_(bf16|f32_mfma|bf16_bpre)(?:_nt)?_kernel.*?(Epi[A-Za-z0-9]+)", name)
    if m:
        tile = re.search(r"ILi(\d+)ELi(\d+)E|<(\d+), (\d+)", name)
        t = "x".join(x for x in (tile.groups() if tile else ()) if x)
        short = "gemm_%s[%s] %s" % (m.group(1), t, m.group(2))
    else:
        short = re.sub(r"^_ZN3dyt\d*|^void dyt::|^dyt::", "", name).split("(")[0][:44]
    print("%-46s calls=%5s avg=%8.1fus tot=%8.2fms %5.1f%%" % (short, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                              float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
print("total kernel time %.2f ms" % (tot / 1e6))
