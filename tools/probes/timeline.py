"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV of the overlapped (multi-stream) run.

usage: timeline.py <kernel_trace.csv> [step index from the end, default 2]
Prints, for the chosen step (adamw_kernel to adamw_kernel): wall time, time with no kernel resident, time-weighted number
of concurrently resident kernels, and the intervals in which every resident kernel is a small one (< 256 workgroups), i.e.
where most CUs have nothing to run."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
    wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1) * max(1, int(r.get("Workgroup_Size_Y", 1) or 1)) * max(1, int(r.get("Workgroup_Size_Z", 1) or 1))
    r["wgs"] = gx // max(1, wx)
    m = re.search(r"(Epi[A-Za-z0-9]+|attn_[a-z_0-9]+|[a-z_0-9]+_kernel)", r["Kernel_Name"])
    r["short"] = (m.group(1) if m else r["Kernel_Name"][:24])
    if "gemm" in r["Kernel_Name"]:
        t = re.search(r"ILi(\d+)ELi(\d+)", r["Kernel_Name"])
        r["short"] = ("bpre" if "bpre" in r["Kernel_Name"] else (t.group(1) if t else "")) + ":" + r["short"]
rows.sort(key=lambda r: r["s"])
marks = [r["e"] for r in rows if "adamw_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-back - 1], marks[-back]
step = [r for r in rows if r["s"] >= t0 and r["e"] <= t1 + 1]
print("step wall %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
ev = []
for r in step:
    ev.append((r["s"], 1, r))
    ev.append((r["e"], -1, r))
ev.sort(key=lambda x: (x[0], x[1]))
live = []
last = t0
idle = 0
conc = 0.0
small = []   # (start, dur, names)
for t, d, r in ev:
    dt = t - last
    if dt > 0:
        if not live:
            idle += dt
        conc += dt * len(live)
        if live and all(x["wgs"] < 256 for x in live):
            small.append((last - t0, dt, ",".join("%s[%d]" % (x["short"], x["wgs"]) for x in live)))
    last = t
    if d == 1:
        live.append(r)
    else:
        live.remove(r)
print("no kernel resident: %.3f ms; mean resident kernels: %.2f" % (idle / 1e6, conc / (t1 - t0)))
tot_small = sum(x[1] for x in small)
print("only small (<256 WG) kernels resident: %.3f ms in %d intervals" % (tot_small / 1e6, len(small)))
agg = {}
for s, dt, names in small:
    agg[names] = agg.get(names, 0) + dt
for names, dt in sorted(agg.items(), key=lambda x: -x[1])[:25]:
    print("  %8.1f us  %s" % (dt / 1e3, names))
if len(sys.argv) > 3:   # full listing
    for r in step:
        print("%9.1f +%7.1f us  wgs %5d  %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["wgs"], r["short"]))
