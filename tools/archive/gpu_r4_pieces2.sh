#!/bin/bash
O=gpurun_out; mkdir -p $O
run() { timeout 300 python bench.py --pieces $1 --single-pass --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --steps ${STEPS:-60} --warmup 4 2> $O/pieces_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('$2 pieces $1: %.4f ms/step  %.4g reads/s  md5 %s' % (d['ms_per_step'], d['value'], d['sketch_md5'][:8]))" | tee -a $O/pieces_sweep2.txt; }
for p in 1 2 4; do HULK_BIN_CHAIN=0 run $p nochain; done
run 1 chain; run 2 chain
python tools/two_ctx_overlap.py 2>&1 | tail -1 | tee -a $O/pieces_sweep2.txt
PIECES=2 python tools/two_ctx_overlap.py 2>&1 | tail -1 | tee -a $O/pieces_sweep2.txt
