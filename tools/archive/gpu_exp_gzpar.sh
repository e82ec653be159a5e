#!/bin/bash
# Block size of the block reader (HULK_INGEST_BLOCK) with the worker teams, file -> sketch, 2 M reads: one gzip member and plain.
# Writes gpurun_out/e2e_gzpar5.txt.
mkdir -p gpurun_out
run() { echo "== $*"; env "${@:2}" timeout 40 python tools/ingest_rate.py 2000000 $1 --gpu 2>&1 | grep "^parse\|^sketch_files" | cut -c1-200; }
{
  for b in 8388608 16777216 33554432 67108864; do run --gz HULK_INGEST_BLOCK=$b; done
  for b in 8388608 16777216 33554432 67108864; do run "" HULK_INGEST_BLOCK=$b; done
} > gpurun_out/e2e_gzpar5.txt 2>&1
cat gpurun_out/e2e_gzpar5.txt
