#!/bin/bash
# Round 4: priority of the flush stream when the flush is NOT skipped (bounds off; C3 with decay): lowest (default) / same as the lanes / highest
O=gpurun_out; mkdir -p $O; : > $O/prio_ab.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --no-long 2>> $O/prio.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$LABEL: %.4f ms/step  md5 %s %s' % (d['ms_per_step'], d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/prio_ab.txt; }
for i in 1 2; do
  LABEL="noprune flush lowest run $i" one --no-prune
  LABEL="noprune flush prio 0 run $i" HULK_FLUSH_PRIORITY=0 one --no-prune
  LABEL="noprune flush prio -1 run $i" HULK_FLUSH_PRIORITY=-1 one --no-prune
done
for p in default 0 -1; do
  for lanes in 1 2; do
    if [ $p = default ]; then unset HULK_FLUSH_PRIORITY; else export HULK_FLUSH_PRIORITY=$p; fi
    python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 --lanes $lanes 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('c3 flush prio $p lanes $lanes: %.3e reads/s (%.3f ms per batch)' % (d['reads_per_s'], d['ms']/ (d['reads_timed']/1.6e6)))" | tee -a $O/prio_ab.txt
  done
done
