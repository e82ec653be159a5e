#!/bin/bash
# does the throw-away ramp context change the cold leg (allocations inside its timed region)?
O=gpurun_out; mkdir -p $O; : > $O/cold2.txt
one() { python bench.py "$@" --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline 2>> $O/cold2.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$LABEL: %.4f ms/step long %.4f cold %s create %.4f' % (d['ms_per_step'], d.get('ms_per_step_long') or 0, d.get('cold_seconds_all_runs'), d.get('cold_create_seconds') or 0))" | tee -a $O/cold2.txt; }
for i in 1 2; do
  LABEL="ramp 40 run $i" one
  LABEL="ramp 0 run $i" HULK_BENCH_RAMP_STEPS=0 one
  LABEL="ramp 40, no long run $i" one --no-long
done
