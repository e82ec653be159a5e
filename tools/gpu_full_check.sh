# everything: GPU test suite, bench line, C3-shaped rate
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full
python -m pytest tests -m gpu -q -rf --timeout=900 > gpurun_out/full/pytest.txt 2>&1; tail -8 gpurun_out/full/pytest.txt
python bench.py --no-cpu-baseline --no-e2e > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/full/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','value_unpruned','ms_per_step_unpruned','value_cold','sketch_md5')})
print('k1a', d['roofline']['avg_launch_us'], 'k1b', d['k_jump_bin']['avg_launch_us'], 'scan', d['roofline_cws_scan']['avg_launch_us'])
PY
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c1-170
