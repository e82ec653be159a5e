#!/bin/bash
# repeat the suite's fuzz_shard slice (HULK_SHARD_DEBUG: every rank's view of the verdicts on stderr); stops at the first failure
O=gpurun_out; mkdir -p $O; : > $O/shardbug.txt
for i in $(seq 1 ${N:-14}); do
  HULK_SHARD_DEBUG=1 timeout 300 python tools/fuzz_shard.py 30 16 > $O/shardbug_run.out 2> $O/shardbug_run.err; rc=$?
  echo "run $i rc=$rc $(tail -1 $O/shardbug_run.out | cut -c1-100) late headers: $(grep -c 'stale header copy' $O/shardbug_run.err)" | tee -a $O/shardbug.txt
  grep 'stale header copy' $O/shardbug_run.err | sort | uniq -c | head -5 >> $O/shardbug.txt
  if [ $rc -ne 0 ]; then grep -v amdgpu.ids $O/shardbug_run.err | tail -120 > $O/shardbug_fail.err; grep MISMATCH $O/shardbug_run.out | cut -c1-600 >> $O/shardbug.txt; break; fi
done
