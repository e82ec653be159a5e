#!/bin/bash
# the whole GPU suite, smoke, and the driver's default bench line
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long','value_unpruned','value_cold','ms_per_step_kernels_alone')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
print('c3', {k:v for k,v in (d.get('c3') or {}).items() if k!='workload'})
print('c5', d.get('c5'))
print('e2e', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in (d.get('e2e') or {}).items()})
print('roofline', d['roofline'] and {k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us')})
PY
