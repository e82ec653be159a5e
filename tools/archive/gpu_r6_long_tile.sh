#!/bin/bash
# the tile form of the long-sequence kernels: parity, fuzz with long shapes, timing against the two-pass form (profiling build)
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py::test_long_sequences_against_oracle_and_split_invariance" -x -q -m gpu > $O/r6h_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r6h_tests.txt
FUZZ_SECONDS=150 timeout 300 python tools/fuzz_parity.py 100000 4242 2>&1 | tail -2
for L in ship nolset; do
  if [ $L = nolset ]; then export HULK_LIB=exp HULK_LONG_NO_LSET=1; fi
  python bench.py --no-c3 --no-c5 --no-e2e --no-cpu-baseline --no-cold --single-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); lr = d['long_reads']
for k in ('reads_5kb', 'contigs_500kb'):
    x = lr[k]; print('$L', k, '%.3g bases/s' % x['bases_per_s'], x['kernels_alone']['us'], x['sketch_md5'][:8], x.get("roofline_k_long_tile", {}).get("frac"))
print('$L', lr.get('fasta_file', {}).get('bases_per_s'), d.get('long_reads_error'))
"
done
