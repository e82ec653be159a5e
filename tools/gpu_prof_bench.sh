cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench; rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --single-pass $BENCH_ARGS > $OUT/bench.json 2> $OUT/err.txt
tail -1 $OUT/bench.json | cut -c1-400
python - <<'PY'
import sqlite3,glob,os
for f in glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_bench/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select name,total_calls,total_duration,average from top_kernels"):
        if 'hulk' in r[0] or 'rocclr' in r[0] or 'nccl' in r[0].lower(): print(r[0][:70], r[1], round(r[2]), round(r[3],1))
PY
