#!/bin/bash
# ablation of k_long_emit (profiling build, HULK_K1_DEBUG bits: 1 = the per-sequence set only, 2 = window minimum only, 4 = set + jump hash without the spectrum's atomics)
export HULK_LIB=exp
for D in 0 4 1 2; do
  HULK_K1_DEBUG=$D python bench.py --no-c3 --no-c5 --no-e2e --no-cpu-baseline --no-cold --single-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); lr = d.get('long_reads') or {}
for k in ('reads_5kb', 'contigs_500kb'):
    x = lr.get(k)
    if x: print('debug $D', k, x['kernels_alone']['us'])
print(d.get('long_reads_error'))
"
done
