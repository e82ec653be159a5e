#!/bin/bash
# Round 4: does the number of HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) matter for the context's streams
# (2 work lanes + flush + estimates)?  The contract test first (fresh box), then A/B.
O=gpurun_out; mkdir -p $O; : > $O/hwq_ab.txt
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/hwq_contract.txt 2>&1; echo "contract rc=$?" >> $O/hwq_contract.txt; tail -3 $O/hwq_contract.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline 2>> $O/hwq.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$LABEL: %.4f ms/step  unpruned %s  long %s  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_unpruned'), d.get('ms_per_step_long'), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/hwq_ab.txt; }
for i in 1 2; do
  LABEL="default queues run $i" one
  LABEL="GPU_MAX_HW_QUEUES=8 run $i" GPU_MAX_HW_QUEUES=8 one
  LABEL="GPU_MAX_HW_QUEUES=2 run $i" GPU_MAX_HW_QUEUES=2 one
  LABEL="GPU_MAX_HW_QUEUES=16 run $i" GPU_MAX_HW_QUEUES=16 one
done
for q in 4 8; do
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 2>/dev/null | tail -1 | sed "s/^/c3 default queues: /" | tee -a $O/hwq_ab.txt
GPU_MAX_HW_QUEUES=$q python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 2>/dev/null | tail -1 | sed "s/^/c3 GPU_MAX_HW_QUEUES=$q: /" | tee -a $O/hwq_ab.txt
done
