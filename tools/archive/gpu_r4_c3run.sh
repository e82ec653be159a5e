O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu -k "json_contract or survives" > $O/c3_contract.txt 2>&1; echo "rc=$?" >> $O/c3_contract.txt; tail -3 $O/c3_contract.txt
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c3.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long')}, {k:v for k,v in d.items() if k.endswith('_error')})
c=d['c3']; print({k:v for k,v in c.items() if k not in ('workload','kernels_alone')}); print(c['kernels_alone'])
PY
