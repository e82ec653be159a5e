#!/bin/bash
# round 5, closing run: the whole GPU suite, smoke, the driver's default bench line, a longer fuzz soak
O=gpurun_out; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long','value_unpruned','value_cold','ms_per_step_kernels_alone')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
print('roofline', {k:d['roofline'].get(k) for k in ('achieved','frac','traffic','avg_launch_us')})
print('scan unpruned', {k:(d.get('roofline_cws_scan_unpruned') or {}).get(k) for k in ('achieved','frac','avg_launch_us')})
c3=d.get('c3') or {}; print('c3', c3.get('value'), c3.get('ms_per_batch'))
c5=d.get('c5') or {}; print('c5', {m:(c5[m]['ms_kernel'], c5[m]['ms_end_to_end']) for m in ('weightedjaccard','jaccard') if m in c5}, (c5.get('directory') or {}).get('seconds_total'))
e=d.get('e2e') or {}; print('e2e', {k:round(v.get('value')/1e6,1) for k,v in e.items() if isinstance(v,dict)})
PY
bash tools/gpu_soak.sh 9800 4 | tail -22
FUZZ_SECONDS=200 timeout 300 python tools/fuzz_devparse.py 100000 9810 2>&1 | tail -1 | tee -a gpurun_out/soak/soak_9800.txt
