# PMC passes for the bench (separate runs, --kernel-trace only): HBM traffic + VALU activity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --single-pass --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --single-pass --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --single-pass --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --single-pass --steps 5 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections,os
OUT=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc'
for f in sorted(glob.glob(OUT+'/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name']
        if 'hulk' not in k: continue
        import re
        k=re.search(r'(k_\w+)',k).group(1)
        agg[k][row['Counter_Name']]+=float(row['Counter_Value']); n[(k,row['Counter_Name'])]+=1
    for k,v in sorted(agg.items()):
        print(k, {a:(round(b/max(n[(k,a)],1),1), n[(k,a)]) for a,b in v.items()})
PY
