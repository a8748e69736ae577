#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 kernel trace: busy time, idle gaps, time with >= 2 kernels in flight."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t_last = ev[0][0]
depth = 0
hist = {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - t_last)
    depth += d
    t_last = t
tot = sum(hist.values())
print("span %.1f ms over %d kernels" % (tot / 1e6, len(rows)))
for k in sorted(hist):
    print("  %d kernels in flight: %8.2f ms  %5.1f%%" % (k, hist[k] / 1e6, 100.0 * hist[k] / tot))
# last N steps only: crude -- use the final 40 % of the span
cut = ev[0][0] + int(0.6 * (ev[-1][0] - ev[0][0]))
depth = 0; t_last = ev[0][0]; h2 = {}
for t, d in ev:
    if t_last >= cut:
        h2[depth] = h2.get(depth, 0) + (t - t_last)
    depth += d; t_last = t
tot2 = sum(h2.values())
print("steady state (last 40%% of span, %.1f ms):" % (tot2 / 1e6))
for k in sorted(h2):
    print("  %d kernels in flight: %8.2f ms  %5.1f%%" % (k, h2[k] / 1e6, 100.0 * h2[k] / tot2))
