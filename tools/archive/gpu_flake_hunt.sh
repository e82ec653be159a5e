#!/bin/bash
# repeat one test until it fails; keep the failing run's output
O=gpurun_out; mkdir -p $O; : > $O/flake.txt
for i in $(seq 1 ${N:-12}); do
  timeout 600 python -m pytest "$@" -x -q > $O/flake_run.txt 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $O/flake_run.txt | cut -c1-120)" | tee -a $O/flake.txt
  if [ $rc -ne 0 ]; then grep -v amdgpu.ids $O/flake_run.txt | tail -150 > $O/flake_fail.txt; break; fi
done
