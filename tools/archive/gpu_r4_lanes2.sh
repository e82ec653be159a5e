#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes or ring_wraparound or bin_then_flush or n_bases or random_reads or k31 or histogram_hook or long_seq" > $O/lanes_parity.txt 2>&1; echo "parity rc=$?" >> $O/lanes_parity.txt; tail -3 $O/lanes_parity.txt
python -c "
import ctypes
from hulk_amd import _lib
" 
for prio in default -1 0 1; do
  if [ $prio = default ]; then unset HULK_FLUSH_PRIORITY; else export HULK_FLUSH_PRIORITY=$prio; fi
  for p in 2; do
    timeout 300 python bench.py --lanes $p --single-pass --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --steps ${STEPS:-60} --warmup 4 2> $O/lanes_$p.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('flush prio $prio lanes $p: %.4f ms/step  %.4g reads/s  md5 %s' % (d['ms_per_step'], d['value'], d['sketch_md5'][:8]))" | tee -a $O/lanes_sweep2.txt
    python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 --lanes $p | cut -c1-150 | tee -a $O/lanes_sweep2.txt
  done
done
