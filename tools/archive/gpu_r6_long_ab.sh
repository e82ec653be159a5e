#!/bin/bash
# long-sequence leg: the shipping library (LDS-set form for long reads) vs every sequence through the HBM-set form (profiling build, HULK_LONG_NO_LSET)
cat > /tmp/lr_print.py <<'PY'
import sys, json
tag = sys.argv[1]
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); lr = d['long_reads']
for k in ('reads_5kb', 'contigs_500kb'):
    x = lr[k]; print(tag, k, '%.3g bases/s' % x['bases_per_s'], x['kernels_alone']['us'], x['sketch_md5'][:8])
print(tag, 'fasta', lr.get('fasta_file', {}).get('bases_per_s'), d.get('long_reads_error'))
PY
for L in ship nolset; do
  if [ $L = nolset ]; then export HULK_LIB=exp HULK_LONG_NO_LSET=1; fi
  python bench.py --no-c3 --no-c5 --no-e2e --no-cpu-baseline --no-cold --single-pass 2>/dev/null | python /tmp/lr_print.py $L
done
