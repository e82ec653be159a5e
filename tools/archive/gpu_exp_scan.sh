cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "pruning or random_reads or drift or c1" 2>&1 | tail -2
for a in "" "--no-prune"; do HULK_NO_OVERLAP=1 python bench.py --no-cpu-baseline --no-cold --single-pass $a 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], 'scan', d['roofline_cws_scan']['avg_launch_us'], d['roofline_cws_scan']['achieved'])"; done
python bench.py --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['value_unpruned'])"
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c1-200
HULK_NO_PRUNE=1 python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 8000000 --interval 100000 --batch 16 | cut -c1-200
