#!/bin/bash
# Round 4: capped k_cws_scan grid (workgroups walk their units) A/B + the nt4-table test of the short-read kernel; parity first
O=gpurun_out; mkdir -p $O; : > $O/scancap_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "nt4 or n_bases or lowercase or lanes or random_reads or pruning or fixture or golden or large_and_ragged or host_chunk or c2_prefix or external_cws or two_groups" > $O/scancap_parity.txt 2>&1; echo "parity rc=$?" >> $O/scancap_parity.txt; tail -3 $O/scancap_parity.txt
one() { python bench.py "$@" --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline 2>> $O/scancap.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d.get('roofline_cws_scan') or {}
print('$LABEL: %.4f ms/step  unpruned %s ms/step  long %s  scan %.1f us (%.0f GB/s, frac %.3f)  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_unpruned'), d.get('ms_per_step_long'), s.get('avg_launch_us',0), s.get('achieved',0), s.get('frac',0), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/scancap_ab.txt; }
for g in 1073741824 4096 2048 8192; do
  LABEL="noprune grid $g" HULK_SCAN_GRID=$g one --no-prune --no-long
done
for i in 1 2 3; do
  for g in 1073741824 4096 2048; do LABEL="default grid $g run $i" HULK_SCAN_GRID=$g one; done
done
timeout 600 python tools/fuzz_parity.py 400 9900 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/scancap_ab.txt
