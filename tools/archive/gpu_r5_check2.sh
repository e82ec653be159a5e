#!/bin/bash
# round 5, call 8: the whole suite again (count-min replay rewritten, scan brackets), the default bench line, unpruned kernel stats
O=gpurun_out; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long','value_unpruned','value_cold','ms_per_step_kernels_alone')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
print('scan unpruned', d.get('roofline_cws_scan_unpruned'))
c3=d.get('c3') or {}
print('c3', c3.get('value'), c3.get('ms_per_batch'), c3.get('kernels_alone'))
c5=d.get('c5') or {}
print('c5', {m:(c5[m]['ms_kernel'], c5[m]['ms_end_to_end'], c5[m]['valu_frac']) for m in ('weightedjaccard','jaccard') if m in c5}, c5.get('directory'))
e=d.get('e2e') or {}
print('e2e', {k:round(v.get('value')/1e6,1) for k,v in e.items() if isinstance(v,dict)})
print('cpu', d.get('cpu_baseline'))
PY
export HULK_LIB=exp R=r05
TITLE="Round 5: each kernel alone (HULK_NO_OVERLAP=1), CWS-scan bounds OFF (bench.py --no-prune = HULK_FLAG_NO_PRUNE on the timed context)" CMD="HULK_LIB=exp HULK_NO_OVERLAP=1 python bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-prune"
HULK_NO_OVERLAP=1 BENCH_ARGS="--no-prune" bash tools/gpu_prof_bench.sh > $O/np.txt 2>&1
python tools/rocprof_summary.py gpurun_out/prof_bench/b_results.db $O/r05_kernel_stats_serial_noprune.md "$TITLE" "$CMD" > /dev/null
grep -E "^\| k_(cms|cws_scan|count|flush|rcp)" $O/r05_kernel_stats_serial_noprune.md | head
