# N-containing reads: parity of the deferred-read path + the bench variant (1 % / 5 % of the reads carry an N)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x > gpurun_out/r02c/pytest.txt 2>&1; tail -3 gpurun_out/r02c/pytest.txt
python tools/fuzz_parity.py 60 > gpurun_out/r02c/fuzz.txt 2>&1; tail -2 gpurun_out/r02c/fuzz.txt
B="python bench.py --no-cpu-baseline --no-cold --single-pass"
$B > gpurun_out/r02c/clean.json 2>/dev/null
for f in 0.01 0.05 0.25; do $B --n-frac $f > gpurun_out/r02c/n_$f.json 2>/dev/null; done
HULK_NO_FAST_K1=1 $B > gpurun_out/r02c/generic_only.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02c/*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value']/1e9,4), round(d['ms_per_step'],4), 'k1a', round(d['roofline']['avg_launch_us'],1), 'k1b', round(d['k_jump_bin']['avg_launch_us'],1), d['sketch_md5'][:8])
PY
