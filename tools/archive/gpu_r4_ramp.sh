#!/bin/bash
# Round 4: discarded steps on a throw-away context right before every pass's warm-up (bench.py RAMP_STEPS) A/B
O=gpurun_out; mkdir -p $O; : > $O/ramp_ab.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline 2>> $O/ramp.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$LABEL: %.4f ms/step  unpruned %.4f  long %.4f  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_unpruned') or 0, d.get('ms_per_step_long') or 0, d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/ramp_ab.txt; }
for i in 1 2 3; do
  LABEL="ramp 0 run $i" HULK_BENCH_RAMP_STEPS=0 one --steps 20 --warmup 5
  LABEL="ramp 40 run $i" HULK_BENCH_RAMP_STEPS=40 one --steps 20 --warmup 5
  LABEL="ramp 100 run $i" HULK_BENCH_RAMP_STEPS=100 one --steps 20 --warmup 5
done
bash tools/gpu_r4_prio.sh
