#!/bin/bash
# Host-ingest knob sweep on the GPU box (parse-only and file -> sketch): readers x parser threads x block size.
# usage (through gpurun): bash tools/gpu_exp_ingest.sh [n_reads]
N=${1:-8000000}
mkdir -p gpurun_out/ingest
OUT=gpurun_out/ingest/sweep.txt
: > $OUT
nproc >> $OUT
python tools/ingest_rate.py $N --gpu >> $OUT 2>&1     # writes the file, default knobs
for R in 2 8 16; do
  echo "== readers $R" >> $OUT
  HULK_INGEST_READERS=$R python tools/ingest_rate.py $N --gpu 2>&1 | tail -2 >> $OUT
done
for T in 8 32 64; do
  echo "== threads $T (readers 8)" >> $OUT
  HULK_INGEST_READERS=8 python tools/ingest_rate.py $N --gpu --threads $T 2>&1 | tail -2 >> $OUT
done
for B in 8388608 16777216 67108864; do
  echo "== block $B (readers 8)" >> $OUT
  HULK_INGEST_READERS=8 HULK_INGEST_BLOCK=$B python tools/ingest_rate.py $N --gpu 2>&1 | tail -2 >> $OUT
done
cat $OUT
