"""Per-(kernel, grid) average duration from a rocprofv3 --kernel-trace CSV of a SERIAL run (DYT_NO_OVERLAP=1): the grid size
separates the shapes a GEMM kernel is launched with.  usage: shape_times.py <kernel_trace.csv> [min total ms]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
agg = {}
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"gemm_(bf16_bpre|bf16_nt|f32_mfma_nt)_kernel.*?(Epi[A-Za-z0-9]+)", n)
    if m:
        t = re.search(r"ILi(\d+)ELi(\d+)", n)
        short = "%s%s %s" % (m.group(1).replace("bf16_", "").replace("_nt", ""), ("[%sx%s]" % t.groups()) if t else "", m.group(2)[:22])
    else:
        mm = re.search(r"(attn_[a-z_0-9]+|[a-z_0-9]+_kernel)", n)
        short = mm.group(1) if mm else n[:30]
    wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    k = (short, wgs)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0, 1e9])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d)
tot = sum(a[1] for a in agg.values())
for (short, wgs), (c, t, mn) in sorted(agg.items(), key=lambda x: -x[1][1]):
    if t / 1e3 < thr:
        continue
    print("%-44s wgs=%5d calls=%5d avg=%8.1f us min=%8.1f us tot=%7.2f ms %4.1f%%" % (short, wgs, c, t / c, mn, t / 1e3, 100 * t / tot))
