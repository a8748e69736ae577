#!/usr/bin/env python3
"""MFMA utilisation per kernel family from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) x 256 CUs x 4 SIMDs)  (the gfx94x derived-counter
formula; ROCm 7.2 has no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots").  SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over SIMDs (16 per v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs
(per-dispatch value / kernel duration = 8 x ~2.1 GHz on this box).
Usage: pmc_mfma.py <pmc_counter_collection.csv>
"""
import csv
import json
import re
import sys

acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    fam = "gemm_bf16 (nt + bpre kernels)" if "gemm_bf16_" in k else ("attn_" + re.search(r"attn_(\w+?)_bf16", k).group(1) if "attn_" in k and "bf16" in k else None)
    if fam is None:
        continue
    d = acc.setdefault(fam, {})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["_n"] = d.get("_n", 0) + 1
out = {}
for fam, d in acc.items():
    busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
    ncounters = len([c for c in d if not c.startswith("_")])
    out[fam] = {"dispatch_rows": d["_n"] // max(ncounters, 1), "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": act,
                "mfma_util": busy / (act / 8 * 256 * 4) if act else None}
print(json.dumps(out, indent=1))
