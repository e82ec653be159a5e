#!/bin/bash
# A/B on ONE box: the end-to-end file figures with the process's region pool (default) and without (HULK_NO_REGION_POOL, profiling build)
# -> profiles/r06_region_pool.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
out=gpurun_out/r06_region_pool.txt; : > $out
for rep in 1 2; do
  for mode in pool nopool; do
    if [ $mode = nopool ]; then export HULK_NO_REGION_POOL=1; else unset HULK_NO_REGION_POOL; fi
    echo "== $mode (run $rep)" >> $out
    HULK_LIB=exp timeout 300 python - >> $out 2>&1 <<'PY'
import json, bench
e = bench.e2e_file_rates()
for k in ("plain", "gz", "bgzf", "plain_8m", "plain_8m_host_parser"):
    print("%-22s %.3g reads/s   runs (s): %s   parse only: %s" % (k, e[k]["value"], " ".join("%.4f" % s for s in e[k]["seconds_all_runs"]),
          ("%.3g" % e[k]["parse_only_reads_per_s"]) if e[k].get("parse_only_reads_per_s") else "-"))
PY
  done
done
cat $out
