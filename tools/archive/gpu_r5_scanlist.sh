#!/bin/bash
# round 5, call 4: list-form CWS scan + double-buffered k_smash: parity, then the default bench line, then C3 with one / two lanes
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_smash.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q > $O/gpu_tests_sl.txt 2>&1; echo "rc=$?" >> $O/gpu_tests_sl.txt; tail -6 $O/gpu_tests_sl.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline > $O/bench_sl.json 2> $O/bench_sl.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_sl.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long','value_unpruned','value_cold','ms_per_step_kernels_alone')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
c3=d.get('c3') or {}
print('c3', c3.get('value'), c3.get('ms_per_batch'), c3.get('kernels_alone'))
c5=d.get('c5') or {}
print('c5', {m:(c5[m]['ms_kernel'], c5[m]['ms_end_to_end'], c5[m]['valu_frac']) for m in ('weightedjaccard','jaccard') if m in c5})
e=d.get('e2e') or {}
print('e2e', {k:round(v.get('value')/1e6,1) for k,v in e.items() if isinstance(v,dict)})
PY
for lanes in 1 2; do
  python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 --lanes $lanes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 lanes $lanes: %.3e reads/s (%.3f ms per batch)' % (d['reads_per_s'], d['ms']/(d['reads_timed']/1.6e6)))"
done
HULK_LIB=exp HULK_SCAN_GRID=1 python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 grid-form scan (profiling build): %.3e reads/s' % d['reads_per_s'])"
