#!/bin/bash
# A 2 M-read .gz (one member, level 1) through hulk_sketch_files with variants of the parallel member reader; then the e2e leg
# of bench.py.  Writes gpurun_out/e2e_gzpar.txt.
mkdir -p gpurun_out
export HULK_INGEST_TRACE=1
run() { echo "== $*"; env "$@" timeout 100 python tools/ingest_rate.py 2000000 --gz --gpu 2>&1 | grep -v "calling thread" | tail -4 | cut -c1-400; }
{
  cat /sys/kernel/mm/transparent_hugepage/enabled
  run A=1
  run HULK_GZ_NO_THP=1
  run HULK_GZ_PAR_CHUNK=524288
  run HULK_GZ_THREADS=8
  run HULK_GZ_THREADS=24
  run HULK_GZ_PAR=0
  timeout 150 python -c "import bench, json; print(json.dumps(bench.e2e_file_rates(2000000)))" 2>&1 | grep -v "calling thread" | tail -3
} > gpurun_out/e2e_gzpar.txt 2>&1
tail -c 7000 gpurun_out/e2e_gzpar.txt
