#!/bin/bash
# bench.py's e2e leg (FASTQ file -> sketch: plain, one gzip member, bgzip'd; 2 M reads) and where the calling thread's time
# goes for each container (tools/ingest_rate.py with HULK_INGEST_TRACE).  Writes gpurun_out/e2e_gzpar4.txt.
mkdir -p gpurun_out
run() { echo "== $*"; env "${@:2}" HULK_INGEST_TRACE=1 timeout 60 python tools/ingest_rate.py 2000000 $1 --gpu 2>&1 | grep -v "amdgpu.ids" | tail -5 | cut -c1-330; }
{
  timeout 100 python -c "import bench, json; print(json.dumps(bench.e2e_file_rates(2000000)))" 2>&1 | tail -1
  run --gz A=1
  run "" A=1
  timeout 30 python -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -1
} > gpurun_out/e2e_gzpar4.txt 2>&1
tail -c 5000 gpurun_out/e2e_gzpar4.txt
