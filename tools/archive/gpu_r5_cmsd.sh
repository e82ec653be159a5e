#!/bin/bash
# round 5, call 5: count-min replay through returning LDS atomics (k_cmsd_freq, k_cms_freq): the order check, parity, C3 rates
O=gpurun_out; mkdir -p $O
make -C tools/ubench lds_atomic_order > /dev/null 2>&1; tools/ubench/lds_atomic_order | tee $O/r05_lds_atomic_order.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_two_rank.py -x -q > $O/gpu_tests_cmsd.txt 2>&1; echo "rc=$?" >> $O/gpu_tests_cmsd.txt; tail -6 $O/gpu_tests_cmsd.txt | cut -c1-300
for lanes in 1 2; do
  python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 --lanes $lanes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 lanes $lanes: %.3e reads/s (%.3f ms per batch)' % (d['reads_per_s'], d['ms']/(d['reads_timed']/1.6e6)))"
done
HULK_LIB=exp HULK_CMSD_CHAIN=1 python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 chain-form k_cmsd_freq (profiling build): %.3e reads/s' % d['reads_per_s'])"
R=r05 bash tools/gpu_prof_c3.sh 2>&1 | grep -E "^\| k_(cmsd|cms_|cws_scan|minimizer_fast|jump|nibble|elem|slot|scan|rcp)" | head -24
timeout 600 python bench.py --no-cpu-baseline --no-c5 --no-long-reads --no-e2e > $O/bench_cmsd.json 2> $O/bench_cmsd.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_cmsd.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long','value_unpruned','value_cold','ms_per_step_kernels_alone')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
c3=d.get('c3') or {}
print('c3', c3.get('value'), c3.get('ms_per_batch'), c3.get('kernels_alone'))
PY
