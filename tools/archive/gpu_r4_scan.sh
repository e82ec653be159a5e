#!/bin/bash
# Round 4: merged-interval scan (no drift) A/B, lane stagger, flush priority; parity first
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "lanes or random_reads or pruning or fixture or golden or large_and_ragged or host_chunk or c2_prefix or external_cws or two_groups" > $O/scan_parity.txt 2>&1; echo "parity rc=$?" >> $O/scan_parity.txt; tail -3 $O/scan_parity.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline 2>> $O/scan.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d.get('roofline_cws_scan') or {}
print('$LABEL: %.4f ms/step  unpruned %s ms/step  long %s  scan %.1f us (%.0f GB/s, frac %.3f)  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_unpruned'), d.get('ms_per_step_long'), s.get('avg_launch_us',0), s.get('achieved',0), s.get('frac',0), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/scan_ab.txt; }
LABEL="noprune merged" one --no-prune
LABEL="noprune per-interval" HULK_SCAN_PER_INTERVAL=1 one --no-prune
for i in 1 2 3; do LABEL="default run $i" one --no-long; done
for i in 1 2 3; do LABEL="flush prio -1 run $i" HULK_FLUSH_PRIORITY=-1 one --no-long; done
for i in 1 2; do LABEL="one lane run $i" one --no-long --lanes 1; done
