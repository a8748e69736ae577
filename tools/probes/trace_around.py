import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]
gk = [k for k in rows[0].keys() if "Grid" in k or "grid" in k]
for r in rows[i0 - 6:i0 + 8]:
    n = r["Kernel_Name"]
    m = re.search(r"(Epi[A-Za-z0-9]+|attn_[a-z_0-9]+|[a-z_0-9]+_kernel)", n)
    print("%-28s %s dur %8.1f us" % (m.group(1) if m else n[:28], [r[k] for k in gk][:3], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
