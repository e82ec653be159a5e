#!/bin/bash
# round 6: the long-sequence path.  (1) ablation of the two-pass form's k_long_emit (profiling build, HULK_LONG_TWO_PASS + HULK_K1_DEBUG bits:
# 4 = set + jump hash without the spectrum's atomics, 2 = the window minimum only), (2) the tile kernel against the two-pass form.
# -> profiles/r06_long_path.txt
cat > /tmp/lr_print.py <<'PY'
import sys, json
tag = sys.argv[1]
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); lr = d.get('long_reads') or {}
for k in ('reads_5kb', 'contigs_500kb'):
    x = lr.get(k)
    if x: print('%-34s %-14s %.3g bases/s  kernels alone (us per Gbase): %s  sketch %s' % (tag, k, x['bases_per_s'], x['kernels_alone']['us'], x['sketch_md5'][:8]))
if d.get('long_reads_error'): print(tag, 'ERROR', d['long_reads_error'])
PY
B="python bench.py --no-c3 --no-c5 --no-e2e --no-cpu-baseline --no-cold --single-pass"
echo "# (1) where k_long_emit's time goes (two-pass form; sketches of the ablated runs are wrong by design)"
for D in 0 4 2; do HULK_LIB=exp HULK_LONG_TWO_PASS=1 HULK_K1_DEBUG=$D $B 2>/dev/null | python /tmp/lr_print.py "two-pass, HULK_K1_DEBUG=$D"; done
echo "# (2) the tile kernel (shipping library) against the two-pass form (profiling build)"
$B 2>/dev/null | python /tmp/lr_print.py "k_long_tile (shipping)"
HULK_LIB=exp HULK_LONG_TWO_PASS=1 $B 2>/dev/null | python /tmp/lr_print.py "k_long_hash + k_long_emit"
