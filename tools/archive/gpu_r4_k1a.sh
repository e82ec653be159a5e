#!/bin/bash
# K1a edits: parity (every kernel instance, N bases, pairs, duplicates), a fuzz slice, then the kernel's own duration
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "instance or n_bases or two_groups or duplicate or repetitive or pair_sets or random_reads or lanes or fixture or read_length" > $O/k1a_parity.txt 2>&1; echo "parity rc=$?" >> $O/k1a_parity.txt; tail -3 $O/k1a_parity.txt
timeout 600 python tools/fuzz_parity.py ${FUZZ_N:-1500} 4242 2>&1 | tail -2 | tee $O/k1a_fuzz.txt
FUZZ_WIDE_W=1 FUZZ_BIG_K=1 timeout 600 python tools/fuzz_parity.py 400 4343 2>&1 | tail -1 | tee -a $O/k1a_fuzz.txt
for i in 1 2; do python bench.py --single-pass --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --steps 60 --warmup 4 2>> $O/k1a.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('run $i: %.4f ms/step  kernels alone %.4f  k1a %.1f us  k1b %.1f us  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_kernels_alone',0), r.get('avg_launch_us',0), d['k_jump_bin']['avg_launch_us'], d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/k1a_bench.txt; done
