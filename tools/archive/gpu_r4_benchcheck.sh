#!/bin/bash
# bench.py after a change: its contract tests (world 1 / 2 / 8 on one GPU) and the driver's default line
O=gpurun_out; mkdir -p $O
[ -n "$SKIP_TESTS" ] || { timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/bc_contract.txt 2>&1; echo "rc=$?" >> $O/bc_contract.txt; tail -3 $O/bc_contract.txt; }
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long','value_unpruned','value_cold','ms_per_step_kernels_alone','ramp_steps')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
print('c3', {k:v for k,v in (d.get('c3') or {}).items() if k not in ('workload','kernels_alone')})
print('roofline', d['roofline'] and {k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us')}, 'cpu', d.get('cpu_baseline'))
PY
