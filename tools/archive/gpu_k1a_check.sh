#!/bin/bash
# k_minimizer_fast after an edit: parity + fuzz slice, the bench's headline and kernel durations, VALU instructions per read (one PMC pass)
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -k "not devparse and not multi_rank" > $O/k1a_tests.txt 2>&1; echo "rc=$?" >> $O/k1a_tests.txt; tail -3 $O/k1a_tests.txt | cut -c1-200
for i in 1 2; do python bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --single-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f  kernels alone %.4f  K1a %.1f us  K1b %.1f us' % (d['ms_per_step'], d['ms_per_step_kernels_alone'], d['roofline']['avg_launch_us'], d['k_jump_bin']['avg_launch_us']))"; done
export HULK_LIB=exp TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_k1a; rm -rf $OUT; mkdir -p $OUT
cd /tmp && HULK_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/sq -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --single-pass --steps 5 --warmup 1 > /dev/null 2> $OUT/sq.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, re, collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob('gpurun_out/pmc_k1a/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        m=re.search(r'(k_\w+)', row['Kernel_Name'])
        if m and row['Counter_Name']=='SQ_INSTS_VALU': agg[m.group(1)]+=float(row['Counter_Value']); cnt[m.group(1)]+=1
for k in ('k_minimizer_fast','k_jump_bin'):
    if cnt[k]: print(k, 'VALU wave-instructions per read: %.2f' % (agg[k]/cnt[k]/1.6e6))
PY
