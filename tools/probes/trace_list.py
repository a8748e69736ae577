import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for i, r in enumerate(rows):
    if sys.argv[2] in r["Kernel_Name"]:
        prev = rows[i - 1]["Kernel_Name"]
        import re
        m = re.search(r"(Epi[A-Za-z0-9<>, a-z]+>|attn_[a-z_0-9]+|[a-z_0-9]+_kernel)", prev)
        out.append("%.0f(%s)" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (m.group(1) if m else prev)[:14]))
print(" ".join(out))
