#!/bin/bash
# A/B: positions per thread of the long-sequence kernels (LONG_PPT 8 / 16 / 32; libhulkhip_ppt*.so built by hand with -DHULK_LONG_PPT)
for L in "" ppt16 ppt32; do
  if [ -n "$L" ]; then export HULK_LIB=$PWD/hulk_amd/csrc/libhulkhip_$L.so; else unset HULK_LIB; fi
  python bench.py --no-c3 --no-c5 --no-e2e --no-cpu-baseline --no-cold --single-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); lr = d['long_reads']
for k in ('reads_5kb', 'contigs_500kb'):
    x = lr[k]; print('${L:-ppt8}', k, '%.3g bases/s' % x['bases_per_s'], x['kernels_alone']['us'], x['sketch_md5'][:8])
"
done
