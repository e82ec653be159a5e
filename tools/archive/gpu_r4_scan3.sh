#!/bin/bash
O=gpurun_out; mkdir -p $O; rm -f $O/scan_pipe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_reads or pruning or fixture or golden or large_and_ragged or concept_drift or ring_wraparound" > $O/scan_parity.txt 2>&1; echo "parity (default kernel) rc=$?" | tee -a $O/scan_pipe.txt
HULK_SCAN_PIPE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_reads or pruning or fixture or golden or large_and_ragged or concept_drift or ring_wraparound" > $O/scan_parity_pipe.txt 2>&1; echo "parity (pipe kernel) rc=$?" | tee -a $O/scan_pipe.txt; tail -2 $O/scan_parity_pipe.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --no-long 2>> $O/scan.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d.get('roofline_cws_scan') or {}
print('$LABEL: %.4f ms/step  scan %.1f us (%.0f GB/s, frac %.3f)  md5 %s %s' % (d['ms_per_step'], s.get('avg_launch_us',0), s.get('achieved',0), s.get('frac',0), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/scan_pipe.txt; }
for i in 1 2; do LABEL="noprune default" one --no-prune; LABEL="noprune pipe" HULK_SCAN_PIPE=1 one --no-prune; done
