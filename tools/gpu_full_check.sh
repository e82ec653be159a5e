# everything: GPU test suite, bench line, C3 rate + profile
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full
python -m pytest tests -m gpu -q -rf --timeout=900 > gpurun_out/full/pytest.txt 2>&1; tail -8 gpurun_out/full/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/full/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','value_unpruned','ms_per_step_unpruned','value_cold','sketch_md5')})
print('k1a', d['roofline']['avg_launch_us'], 'k1b', d['k_jump_bin']['avg_launch_us'], 'scan', d['roofline_cws_scan']['avg_launch_us'])
PY
bash tools/gpu_prof_c3.sh 2>&1 | grep -E "reads_per_s|k_cmsd|k_cws_scan|k_scan_test|k_cws_resolve|k_slot_tmin|k_minimizer_fast|k_jump|k_nibble|k_elem|k_count|k_rcp"
