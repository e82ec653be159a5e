#!/bin/bash
# K1c placement: range size / workgroup shape of k_nibble_hist, light recount kernel
O=gpurun_out; mkdir -p $O; rm -f $O/nib_sweep.txt
for cfg in "18 1024" "16 256"; do set -- $cfg
HULK_NIB_RLOG=$1 HULK_NIB_BLOCK=$2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nibble or random_reads or fixture or lanes or k31 or ring_wraparound or repetitive or every_short" > $O/nib_parity.txt 2>&1; echo "parity rlog=$1 block=$2 rc=$? $(tail -1 $O/nib_parity.txt)" | tee -a $O/nib_sweep.txt
done
one() { python bench.py "$@" --single-pass --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-cpu-baseline --steps 60 --warmup 4 2>> $O/nib.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$LABEL: %.4f ms/step  kernels alone %.4f  md5 %s %s' % (d['ms_per_step'], d.get('ms_per_step_kernels_alone',0), d['sketch_md5'][:8], [k for k in d if k.endswith('_error')]))" | tee -a $O/nib_sweep.txt; }
for cfg in "18 1024" "17 512" "16 256" "16 1024" "15 256" "18 1024" "16 256"; do set -- $cfg
  LABEL="lanes 2 rlog $1 block $2" HULK_NIB_RLOG=$1 HULK_NIB_BLOCK=$2 one
done
LABEL="lanes 1 rlog 18 block 1024" HULK_NIB_RLOG=18 HULK_NIB_BLOCK=1024 one --lanes 1
LABEL="lanes 1 rlog 16 block 256" HULK_NIB_RLOG=16 HULK_NIB_BLOCK=256 one --lanes 1
HULK_NIB_RLOG=16 HULK_NIB_BLOCK=256 python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c1-150 | tee -a $O/nib_sweep.txt
HULK_NIB_RLOG=18 HULK_NIB_BLOCK=1024 python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c1-150 | tee -a $O/nib_sweep.txt
