cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
for d in 64 32 24 8 0; do
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
OUT=/tmp/k1p; rm -rf $OUT
cd /tmp && HULK_K1_DEBUG=$d rocprofv3 --kernel-trace --pmc $set -d $OUT -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/k1_ablate.py child > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for f in glob.glob('/tmp/k1p/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(float); n=collections.Counter()
    for row in csv.DictReader(open(f)):
        if 'k_minimizer_fast' in row['Kernel_Name']:
            agg[row['Counter_Name']]+=float(row['Counter_Value']); n[row['Counter_Name']]+=1
    print('dbg=$d', {a:round(b/max(n[a],1)/1e6,2) for a,b in agg.items()}, '(millions per launch)')
PY
done; done
