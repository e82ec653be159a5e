#!/bin/bash
# Round 4: work lanes (one per spectrum ring, free-running on the private streams).  Parity subset, then the C2 step with 1 / 2 lanes.
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes or ring_wraparound or bin_then_flush or n_bases or random_reads or k31 or histogram_hook or long_seq" > $O/lanes_parity.txt 2>&1; echo "parity rc=$?" >> $O/lanes_parity.txt; tail -3 $O/lanes_parity.txt
for p in 1 2 2 1; do
  timeout 300 python bench.py --lanes $p --single-pass --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --steps ${STEPS:-60} --warmup 4 2> $O/lanes_$p.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('lanes $p: %.4f ms/step  %.4g reads/s  kernels alone %.4f ms/step  k1a %.1f us  md5 %s  errors %s' % (d['ms_per_step'], d['value'], d.get('ms_per_step_kernels_alone', 0), r.get('avg_launch_us', 0), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/lanes_sweep.txt
done
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 --lanes 1 | cut -c1-200 | tee -a $O/lanes_sweep.txt
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 --lanes 2 | cut -c1-200 | tee -a $O/lanes_sweep.txt
