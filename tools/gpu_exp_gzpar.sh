#!/bin/bash
# Where the calling thread's time goes, file -> sketch, 2 M reads: one gzip member (parallel reader), bgzip'd, plain; parser
# threads and block size varied for the .gz.  Writes gpurun_out/e2e_gzpar3.txt.
mkdir -p gpurun_out
export HULK_INGEST_TRACE=1
run() { echo "== $*"; env "${@:2}" timeout 100 python tools/ingest_rate.py 2000000 $1 --gpu 2>&1 | grep -v "amdgpu.ids" | tail -5 | cut -c1-330; }
{
  run --gz A=1
  run "--gz --threads 8" A=1
  run "--gz --threads 32" A=1
  run --gz HULK_INGEST_BLOCK=16777216
  run --gz HULK_INGEST_BLOCK=67108864
  run --bgzf A=1
  run "" A=1
} > gpurun_out/e2e_gzpar3.txt 2>&1
tail -c 9000 gpurun_out/e2e_gzpar3.txt
