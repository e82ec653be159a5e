#!/bin/bash
O=gpurun_out; mkdir -p $O; : > $O/twoctx2.txt
for q in 4 8 16; do
echo "== first use allocates, GPU_MAX_HW_QUEUES=$q" | tee -a $O/twoctx2.txt
GPU_MAX_HW_QUEUES=$q HULK_NO_PRERESERVE=1 python tools/two_ctx_idle.py 2>&1 | grep -v amdgpu.ids | grep -E "fresh|^D|B with" | tee -a $O/twoctx2.txt
done
echo "== first use allocates, one lane" | tee -a $O/twoctx2.txt
HULK_WORK_LANES=1 HULK_NO_PRERESERVE=1 python tools/two_ctx_idle.py 2>&1 | grep -v amdgpu.ids | grep -E "fresh|^D|B with" | tee -a $O/twoctx2.txt
