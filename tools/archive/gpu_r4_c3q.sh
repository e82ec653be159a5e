#!/bin/bash
O=gpurun_out; mkdir -p $O; : > $O/c3q.txt
for q in 4 16; do for lanes in 1 2; do
GPU_MAX_HW_QUEUES=$q python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 --lanes $lanes 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('c3 GPU_MAX_HW_QUEUES=$q lanes $lanes: %.3e reads/s (%.3f ms per batch)' % (d['reads_per_s'], d['ms']/ (d['reads_timed']/1.6e6)))" | tee -a $O/c3q.txt
done; done
