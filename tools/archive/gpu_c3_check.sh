# drift-mode parity (count-min with decay, drift resolve) + the C3-shaped rate and per-kernel profile
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "drift or c3" > gpurun_out/c3/pytest.txt 2>&1; tail -3 gpurun_out/c3/pytest.txt
bash tools/gpu_prof_c3.sh
