#!/bin/bash
O=gpurun_out; mkdir -p $O; : > $O/twoctx.txt
echo "== lanes' buffers reserved by hulk_create" | tee -a $O/twoctx.txt
python tools/two_ctx_idle.py 2>&1 | grep -v amdgpu.ids | tee -a $O/twoctx.txt
echo "== HULK_NO_PRERESERVE=1 (first use allocates)" | tee -a $O/twoctx.txt
HULK_NO_PRERESERVE=1 python tools/two_ctx_idle.py 2>&1 | grep -v amdgpu.ids | tee -a $O/twoctx.txt
echo "== one work lane, reserved" | tee -a $O/twoctx.txt
HULK_WORK_LANES=1 python tools/two_ctx_idle.py 2>&1 | grep -v amdgpu.ids | tee -a $O/twoctx.txt
sed -i 's/HULK_BENCH_RAMP_STEPS=0/HULK_BENCH_RAMP_MS=0/' tools/gpu_r4_cold2.sh
bash tools/gpu_r4_cold2.sh
