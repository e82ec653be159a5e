#!/bin/bash
# lane buffers reserved by hulk_create: parity subset, then the cold leg with / without the ramp context
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cpp_host.py -x -q -m gpu > $O/pre_parity.txt 2>&1; echo "parity rc=$?" >> $O/pre_parity.txt; tail -3 $O/pre_parity.txt
bash tools/gpu_r4_cold2.sh
HULK_NO_PRERESERVE=1 python bench.py --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --no-long 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('no prereserve: cold', d.get('cold_seconds_all_runs'), 'create', d.get('cold_create_seconds'))" | tee -a $O/cold2.txt
