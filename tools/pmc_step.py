#!/usr/bin/env python3
"""Per-kernel-family HBM traffic and MFMA utilisation of the whole fine-tune step from the three rocprofv3 --pmc passes of
tools/pmc_step.sh.  FETCH_SIZE / WRITE_SIZE are KiB of the L2's memory-side requests; on gfx950 FETCH_SIZE is DOUBLED (128-B requests
tallied at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE is taken as reported.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES /
((GRBM_GUI_ACTIVE / 8 XCDs) x 256 CUs x 4 SIMDs).
usage: pmc_step.py <steps> <fetch.csv> <write.csv> <mfma.csv> <kernel_trace.csv> <traffic.json> <mfma.json>"""
import csv, json, re, sys

steps = int(sys.argv[1])


def family(k):
    k = k.split("(")[0]
    m = re.search(r"gemm_bf16_(nt|bpre)_kernel", k)
    if m:
        epi = re.search(r"Epi[A-Za-z0-9]+", k)
        tile = re.search(r"ILi(\d+)ELi(\d+)E", k)
        return "gemm %s%s %s" % (m.group(1), " %sx%s" % tile.groups() if tile and m.group(1) == "nt" else "", epi.group(0) if epi else "")
    for n in ("attn_fwd", "attn_bwd_fused", "attn_bwd_dq", "attn_bwd_dkv", "wgrad_bf16", "wgrad_reduce", "tok_bwd", "ln_bwd", "ln_fwd", "ln_gather", "gate_logits",
              "gate_select", "bwd_prep", "reduce_partials", "head_", "loss_", "adamw", "im2col", "prep_adapters", "fillBuffer", "copyBuffer"):
        if n in k:
            return n.rstrip("_")
    return "other"


excluded_GB = 0.0   # the context's one-time arena memset (15 GB, dyt_ctx_create) is not part of a step


def collect(path, counters):
    global excluded_GB
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in counters:
            continue
        if "fillBuffer" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE" and float(r["Counter_Value"]) * 1024 > 1e9:
            excluded_GB += float(r["Counter_Value"]) * 1024 / 1e9
            continue
        d = acc.setdefault(family(r["Kernel_Name"]), {"n": 0})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r["Counter_Name"] == counters[0]:
            d["n"] += 1
    return acc


fetch, write = collect(sys.argv[2], ["FETCH_SIZE"]), collect(sys.argv[3], ["WRITE_SIZE"])
rows, tot = {}, 0.0
for fam in sorted(set(fetch) | set(write)):
    f = 2.0 * fetch.get(fam, {}).get("FETCH_SIZE", 0.0) * 1024
    w = write.get(fam, {}).get("WRITE_SIZE", 0.0) * 1024
    n = fetch.get(fam, write.get(fam))["n"]
    rows[fam] = {"launches_per_step": round(n / steps, 1), "fetch_GB_per_step": round(f / steps / 1e9, 3), "write_GB_per_step": round(w / steps / 1e9, 3),
                 "MB_per_launch": round((f + w) / max(n, 1) / 1e6, 1)}
    tot += (f + w) / steps
# kernel durations of the same profiled run (the MFMA pass's kernel trace)
dur = {}
try:
    for r in csv.DictReader(open(sys.argv[5])):
        fam = family(r["Kernel_Name"])
        dur[fam] = dur.get(fam, 0.0) + (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
except Exception as e:  # noqa
    dur = {}
for fam, ms in dur.items():
    if fam in rows:
        rows[fam]["ms_per_step_under_the_profiler"] = round(ms / steps, 3)
        rows[fam]["TB_per_s"] = round((rows[fam]["fetch_GB_per_step"] + rows[fam]["write_GB_per_step"]) / max(ms / steps, 1e-9), 2)
out = {"steps_profiled": steps, "hbm_GB_per_step": round(tot / 1e9, 2), "kernel_ms_per_step_under_the_profiler": round(sum(dur.values()) / steps, 2),
       "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported; KiB -> bytes; separate --pmc passes",
       "excluded_one_time_arena_memset_GB": round(excluded_GB, 2), "families": dict(sorted(rows.items(), key=lambda kv: -(kv[1]["fetch_GB_per_step"] + kv[1]["write_GB_per_step"])))}
json.dump(out, open(sys.argv[6], "w"), indent=1)
mf = collect(sys.argv[4], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
res, tb, ta = {}, 0.0, 0.0
for fam, d in mf.items():
    busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
    tb += busy; ta += act
    if busy > 0:
        res[fam] = {"launches_per_step": round(d["n"] / steps, 1), "mfma_util": round(busy / (act / 8 * 256 * 4), 4) if act else None,
                    "share_of_gpu_active_cycles": None}
for fam, d in mf.items():
    if fam in res:
        res[fam]["share_of_gpu_active_cycles"] = round(d.get("GRBM_GUI_ACTIVE", 0.0) / ta, 4)
json.dump({"steps_profiled": steps, "whole_step_mfma_util": round(tb / (ta / 8 * 256 * 4), 4), "formula": "SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 * 4)",
           "families": dict(sorted(res.items(), key=lambda kv: -kv[1]["share_of_gpu_active_cycles"]))}, open(sys.argv[7], "w"), indent=1)
print(json.dumps({"hbm_GB_per_step": out["hbm_GB_per_step"], "whole_step_mfma_util": round(tb / (ta / 8 * 256 * 4), 4)}))
