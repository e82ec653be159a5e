#!/bin/bash
# The e2e leg of bench.py alone (FASTQ file -> sketch: plain, one gzip member, bgzip'd), then the .gz file again with the
# parallel member reader off, for the before/after of GzPar on the GPU box's host.  Writes gpurun_out/e2e_gzpar.txt.
mkdir -p gpurun_out
export HULK_INGEST_TRACE=1
{
  timeout 240 python -c "import bench, json; print(json.dumps(bench.e2e_file_rates(2000000)))"
  HULK_GZ_PAR=0 timeout 120 python -c "import bench, json; r = bench.e2e_file_rates(1000000); print('HULK_GZ_PAR=0', json.dumps(r['gz']))"
} > gpurun_out/e2e_gzpar.txt 2>&1
tail -c 3000 gpurun_out/e2e_gzpar.txt
