#!/bin/bash
# The e2e leg of bench.py (FASTQ file -> sketch: plain, one gzip member, bgzip'd), huge pages on and off, and the GPU ingest
# tests.  Writes gpurun_out/e2e_gzpar.txt.
mkdir -p gpurun_out
{
  cat /sys/kernel/mm/transparent_hugepage/enabled
  timeout 100 python -c "import bench, json; print(json.dumps(bench.e2e_file_rates(2000000)))" 2>&1 | tail -1
  echo "== HULK_GZ_NO_THP=1"
  HULK_GZ_NO_THP=1 timeout 100 python -c "import bench, json; print(json.dumps(bench.e2e_file_rates(2000000)))" 2>&1 | tail -1
  timeout 60 python -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -2
} > gpurun_out/e2e_gzpar2.txt 2>&1
tail -c 4000 gpurun_out/e2e_gzpar2.txt
