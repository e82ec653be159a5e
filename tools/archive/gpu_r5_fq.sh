#!/bin/bash
# round 5, call 3: the device FASTQ parser (fuzz against the line pump oracle, the ingest tests, the e2e leg), the register-tiled
# k_smash (parity + C5 leg), the void-header test, the profiling build's smoke
O=gpurun_out; mkdir -p $O
timeout 600 python tools/fuzz_devparse.py 80 11 > $O/fuzz_devparse.txt 2>&1; echo "fuzz_devparse rc=$? $(tail -1 $O/fuzz_devparse.txt)"
grep -m5 MISMATCH $O/fuzz_devparse.txt | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_smash.py tests/test_gpu_two_rank.py tests/test_gpu_cpp_host.py -x -q > $O/gpu_tests_fq.txt 2>&1; echo "rc=$?" >> $O/gpu_tests_fq.txt; tail -15 $O/gpu_tests_fq.txt | cut -c1-300
HULK_LIB=exp HULK_NO_OVERLAP=1 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --no-cpu-baseline --no-cold --no-c3 --single-pass > $O/bench_fq.json 2> $O/bench_fq.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_fq.json'))
print({k: d.get(k) for k in ('value','ms_per_step')})
print('errors', {k:v for k,v in d.items() if k.endswith('_error')})
print('c5', json.dumps(d.get('c5'))[:1500])
e=d.get('e2e') or {}
print('e2e', {k:(round(v.get('value')/1e6,1), v.get('seconds_all_runs')) if isinstance(v,dict) else v for k,v in e.items()})
PY
HULK_INGEST_TRACE=1 python - <<'PY' 2>&1 | grep -v amdgpu | tail -12
import os, time, tempfile, shutil, sys
sys.path.insert(0, os.getcwd())
import torch, hulk_amd
from hulk_amd import synth
d = tempfile.mkdtemp(prefix="hulk_fq_", dir="/dev/shm")
p = os.path.join(d, "r.fq"); n = 2_000_000
qual = b"I" * 150
with open(p, "wb") as fh:
    for first in range(0, n, 100000):
        bases, _ = synth.reads_numpy(first, 100000, 150); bb = bases[:100000*150].tobytes()
        fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (first+i, bb[i*150:(i+1)*150], qual) for i in range(100000)))
big = os.path.join(d, "r4.fq")
with open(big, "wb") as fo:
    for _ in range(4):
        with open(p, "rb") as fi: shutil.copyfileobj(fi, fo, 1 << 24)
for readers in (4, 8, 16):
    for blk in (8 << 20, 16 << 20, 32 << 20):
        ts = []
        for _ in range(3):
            g = hulk_amd.GpuSketcher(21, 9, 512, interval=100000)
            t0 = time.perf_counter(); st = g.sketch_files([big], opts={"file_readers": readers, "block_bytes": blk}); g.finish(); ts.append(time.perf_counter() - t0); g.close()
        print(f"device parser: readers {readers} block {blk>>20} MiB: {8e6/min(ts)/1e6:.1f} M reads/s ({os.path.getsize(big)/min(ts)/1e9:.1f} GB/s) runs {[round(x,3) for x in ts]}", flush=True)
shutil.rmtree(d)
PY
