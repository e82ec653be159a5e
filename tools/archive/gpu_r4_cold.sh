#!/bin/bash
O=gpurun_out; mkdir -p $O
for lanes in 0 1 0; do
HULK_BENCH_PREWARM_S=0 HULK_BENCH_LONG_STEPS=6 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c3 --no-c5 --no-long-reads --no-e2e --lanes $lanes 2>$O/cold.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('lanes $lanes', 'cold runs', d.get('cold_seconds_all_runs'), 'create', d.get('cold_create_seconds'), 'ms/step', d['ms_per_step'], 'long', d.get('ms_per_step_long'), [k for k in d if k.endswith('_error')])"
done
