#!/bin/bash
# Where does the first pass of a process lose its ~1.5 ms?  Per-step wall times of every pass (no pre-warm), both ways.
mkdir -p gpurun_out/firstpass; O=gpurun_out/firstpass/steps.txt; : > $O
for M in 1 2; do
  echo "== HULK_BENCH_STEPTIMES=$M" >> $O
  HULK_BENCH_STEPTIMES=$M HULK_BENCH_PREWARM_S=0 HULK_BENCH_REPEAT=2 python bench.py --no-cpu-baseline --no-cold --no-e2e --single-pass 2>&1 | grep -v amdgpu.ids | grep -v '^{' >> $O
done
cat $O
