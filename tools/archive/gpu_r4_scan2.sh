#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_reads or pruning or fixture or golden or large_and_ragged or external_cws or concept_drift" > $O/scan_parity.txt 2>&1; echo "parity rc=$?" >> $O/scan_parity.txt; tail -3 $O/scan_parity.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline 2>> $O/scan.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d.get('roofline_cws_scan') or {}
print('$LABEL: %.4f ms/step  unpruned %s ms/step  scan %.1f us (%.0f GB/s, frac %.3f)  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_unpruned'), s.get('avg_launch_us',0), s.get('achieved',0), s.get('frac',0), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/scan_ab.txt; }
LABEL="noprune merged + group pipeline" one --no-prune
LABEL="noprune per-interval + group pipeline" HULK_SCAN_PER_INTERVAL=1 one --no-prune
LABEL="default" one --no-long
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c1-150 | tee -a $O/scan_ab.txt
