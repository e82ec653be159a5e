#!/bin/bash
# Round 4: the work lanes.  Parity of the pieces against the oracle, then the C2 step with 1 / 2 / 4 / 8 / 16 pieces.
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pieces or ring_wraparound or bin_then_flush or n_bases" > $O/pieces_parity.txt 2>&1; echo "parity rc=$?" >> $O/pieces_parity.txt; tail -3 $O/pieces_parity.txt
for p in 1 2 4 8 16; do
  timeout 300 python bench.py --pieces $p --single-pass --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --steps ${STEPS:-60} --warmup 4 2> $O/pieces_$p.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('pieces $p: %.4f ms/step  %.4g reads/s  kernels alone %.4f ms/step  k1a %.1f us  md5 %s  errors %s' % (d['ms_per_step'], d['value'], d.get('ms_per_step_kernels_alone', 0), r.get('avg_launch_us', 0), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/pieces_sweep.txt
done
