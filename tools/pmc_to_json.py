#!/usr/bin/env python3
"""gpurun_out/pmc/**/counter_collection.csv (tools/gpu_pmc.sh) -> profiles/r01_pmc.json
Per-launch averages per kernel; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024:
FETCH_SIZE/WRITE_SIZE are reported in KiB and on gfx950 FETCH_SIZE reads half of the fetched bytes
(calibrated on k_build_k32: 2.39 GB read -> 1.19e6, 398.3 MB written -> 389120)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r01_pmc.json")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in sorted(glob.glob(src + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", row["Kernel_Name"])
        if "hulk" not in row["Kernel_Name"] or not m:
            continue
        k, c = m.group(1), row["Counter_Name"]
        agg[k][c] += float(row["Counter_Value"]); cnt[(k, c)] += 1
doc = {
    "source": "rocprofv3 --kernel-trace --pmc <set> (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*) -- python bench.py "
              "--no-cpu-baseline --steps 5 --warmup 1  (tools/gpu_pmc.sh, tools/pmc_to_json.py); per-launch averages; "
              "16 intervals = 1.6e6 reads per launch",
    "units": "FETCH_SIZE/WRITE_SIZE as reported (KiB); hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 - gfx950 "
             "FETCH_SIZE reads half of the fetched bytes (calibrated on k_build_k32: 2.39 GB read -> 1.19e6, 398.3 MB written -> 389120)",
}
for k in sorted(agg):
    d = {c: round(v / cnt[(k, c)], 1) for c, v in sorted(agg[k].items())}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    doc[k] = d
json.dump(doc, open(out, "w"), indent=1)
print("wrote", out, "kernels:", len(doc) - 2)
