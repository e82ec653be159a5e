#!/usr/bin/env python3
"""gpurun_out/pmc/**/counter_collection.csv (tools/gpu_pmc.sh) -> profiles/rNN_pmc.json
Per-launch averages per kernel over the launches of the timed region and the warm-up of
`bench.py --no-cpu-baseline --no-cold --single-pass --steps 5 --warmup 1` (16 intervals = 1.6e6 reads per launch).

HBM bytes follow MI355X_MICROARCH.md's rocprofv3 section: FETCH_SIZE / WRITE_SIZE come from separate --pmc passes and are
reported in KiB; on gfx950 FETCH_SIZE reads HALF of the fetched bytes (calibrated here on k_build_k32: 2.39 GB read ->
1.19e6, 398.3 MB written -> 389120), so hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024.

Derived figures (formulas stated so that they can be re-derived from the raw counters in the same file):
  avg_us                    mean kernel duration of the same launches (kernel-trace timestamps of the `sq` pass)
  valu_issue_floor_us       SQ_INSTS_VALU / 1024 SIMDs * 2 cycles / 2.4 GHz (every VALU wave-instruction at full rate)
  valu_issue_frac           valu_issue_floor_us / avg_us
  valu_cycles_frac          SQ_INST_CYCLES_VALU / (1024 SIMDs * avg_us * 2400 cycles/us): cycles the VALU pipes were occupied,
                            as the counter reports them (multi-cycle instructions counted with their issue cycles)
  lds_bank_conflict_frac    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: share of the LDS pipe's active cycles lost to bank conflicts
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_pmc.json")
reads_per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 1_600_000
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in sorted(glob.glob(src + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", row["Kernel_Name"])
        if "hulk" not in row["Kernel_Name"] or not m:
            continue
        k, c = m.group(1), row["Counter_Name"]
        agg[k][c] += float(row["Counter_Value"]); cnt[(k, c)] += 1
dur = collections.defaultdict(list)
for f in sorted(glob.glob(src + "/sq/**/*kernel_trace.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", row["Kernel_Name"])
        if "hulk" in row["Kernel_Name"] and m:
            dur[m.group(1)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
doc = {
    "source": "rocprofv3 --kernel-trace --pmc <set> (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* | VALU | LDS | TCC_* | GRBM) -- "
              "python bench.py --no-cpu-baseline --no-cold --single-pass --steps 5 --warmup 1  (tools/gpu_pmc.sh, "
              "tools/pmc_to_json.py); per-launch averages",
    "reads_per_launch": reads_per_launch,
    "units": "FETCH_SIZE/WRITE_SIZE as reported (KiB); hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 - gfx950 "
             "FETCH_SIZE reads half of the fetched bytes (calibrated on k_build_k32: 2.39 GB read -> 1.19e6, 398.3 MB written -> 389120)",
    "formulas": {"valu_issue_floor_us": "SQ_INSTS_VALU / 1024 * 2 / 2400", "valu_issue_frac": "valu_issue_floor_us / avg_us",
                 "valu_cycles_frac": "SQ_INST_CYCLES_VALU / (1024 * avg_us * 2400)",
                 "lds_bank_conflict_frac": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"},
}
for k in sorted(agg):
    d = {c: round(v / cnt[(k, c)], 1) for c, v in sorted(agg[k].items())}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if dur.get(k):
        d["avg_us"] = round(sum(dur[k]) / len(dur[k]), 2)
        if "SQ_INSTS_VALU" in d:
            d["valu_issue_floor_us"] = round(d["SQ_INSTS_VALU"] / 1024 * 2 / 2400, 2)
            d["valu_issue_frac"] = round(d["valu_issue_floor_us"] / d["avg_us"], 4) if d["avg_us"] else None
        if "SQ_INST_CYCLES_VALU" in d and d["avg_us"]:
            d["valu_cycles_frac"] = round(d["SQ_INST_CYCLES_VALU"] / (1024 * d["avg_us"] * 2400), 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    doc[k] = d
json.dump(doc, open(out, "w"), indent=1)
print("wrote", out, "kernels:", len(doc) - 4)
