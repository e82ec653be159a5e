#!/bin/bash
# Round 4: region capacity of the minimizer list bounded by FAST_CAND per read (1024 + pad entries at w = 9 instead of 2304): parity, then A/B by pad
O=gpurun_out; mkdir -p $O; : > $O/rcap_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/rcap_parity.txt 2>&1; echo "parity rc=$?" >> $O/rcap_parity.txt; tail -3 $O/rcap_parity.txt
timeout 400 python tools/fuzz_parity.py 600 9950 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/rcap_ab.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline 2>> $O/rcap.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$LABEL: %.4f ms/step  unpruned %.4f  long %.4f  alone %.4f  K1a %.1f us  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_unpruned') or 0, d.get('ms_per_step_long') or 0, d.get('ms_per_step_kernels_alone') or 0, d['roofline']['avg_launch_us'], d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/rcap_ab.txt; }
for i in 1 2; do
  for pad in 64 0 128 320 100000; do LABEL="pad $pad run $i" HULK_RCAP_PAD=$pad one; done
done
for pad in 64 100000; do
  echo "== two contexts, first use allocates, pad $pad" | tee -a $O/rcap_ab.txt
  HULK_RCAP_PAD=$pad HULK_NO_PRERESERVE=1 python tools/two_ctx_idle.py 2>&1 | grep -v amdgpu.ids | tee -a $O/rcap_ab.txt
done
