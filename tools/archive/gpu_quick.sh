# quick regression: spectrum/sketch parity subset, C2 bench line, C3-shaped rate
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quick
python -m pytest tests/test_gpu_parity.py -q -x -k "fixture or random_reads or nibble or batches or drift or two_groups" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-cold --no-e2e --single-pass 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('C2', round(d['value']/1e9,4), round(d['ms_per_step'],4), 'k1a', round(d['roofline']['avg_launch_us'],1), 'k1b', round(d['k_jump_bin']['avg_launch_us'],1), d['sketch_md5'][:8])"
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c1-170
