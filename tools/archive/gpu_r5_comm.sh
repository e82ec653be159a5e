#!/bin/bash
# round 5, call 2: the whole GPU suite on the new exchange (one stream, sealed headers) + two-vector scan; the multi-rank fuzz
# at 16 hardware queues with the header-health counters; the reproducer's variants 6 and 7; the default bench line
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
: > $O/r05_shard16.txt
for i in 1 2 3 4 5 6; do
  GPU_MAX_HW_QUEUES=16 timeout 300 python tools/fuzz_shard.py 30 16 > $O/shard16_run.out 2> $O/shard16_run.err; rc=$?
  echo "GPU_MAX_HW_QUEUES=16 run $i rc=$rc $(tail -1 $O/shard16_run.out | cut -c1-200)" | tee -a $O/r05_shard16.txt
  grep MISMATCH $O/shard16_run.out | cut -c1-400 >> $O/r05_shard16.txt
done
make -C tools/ubench hdr_race > /dev/null 2>&1
for q in 4 16; do
  GPU_MAX_HW_QUEUES=$q timeout 200 tools/ubench/hdr_race 20000 8 0 67 >> $O/r05_hdr_race2.txt 2>> $O/r05_hdr_race2.err
done
cat $O/r05_hdr_race2.txt; head -5 $O/r05_hdr_race2.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_long','value_unpruned','value_cold','ms_per_step_kernels_alone')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
print('c3', {k:v for k,v in (d.get('c3') or {}).items() if k!='workload'})
print('scan', d.get('roofline_cws_scan'))
print('roofline', d['roofline'] and {k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us')})
PY
