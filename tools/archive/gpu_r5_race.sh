#!/bin/bash
# round 5, call 1: the header-ordering reproducer over hardware-queue counts, ranks and load; RCCL world 2 on one GPU
O=gpurun_out; mkdir -p $O; : > $O/r05_hdr_race.txt
make -C tools/ubench hdr_race > /dev/null 2>&1
for q in 4 16; do
  for cfg in "1 0" "4 0" "8 0" "8 1"; do
    set -- $cfg
    GPU_MAX_HW_QUEUES=$q timeout 300 tools/ubench/hdr_race ${ITERS:-60000} $1 $2 12345 >> $O/r05_hdr_race.txt 2>> $O/r05_hdr_race.err
  done
done
timeout 200 python tools/rccl_world2_one_gpu.py > $O/r05_rccl_world2.txt 2> $O/r05_rccl_world2.err
tail -3 $O/r05_rccl_world2.txt
grep -i "nccl\|rccl" $O/r05_rccl_world2.err | grep -iv "amdgpu.ids" | head -10
cat $O/r05_hdr_race.txt
head -20 $O/r05_hdr_race.err
