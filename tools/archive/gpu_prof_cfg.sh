# rocprofv3 kernel stats of tools/run_config.py with $CFG_ARGS (e.g. "--len 10000 --reads 100000 --interval 2500 --batch 8")
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg; rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o b -- python $GRAFT_REPO_ROOT/tools/run_config.py --k 21 --S 512 --decay 1.0 $CFG_ARGS > $OUT/out.json 2> $OUT/err.txt
tail -1 $OUT/out.json | cut -c1-200
python - <<'PY'
import sqlite3,glob,os
for f in glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_cfg/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select name,total_calls,total_duration,average from top_kernels"):
        if 'hulk' in r[0]: print(r[0][:70], r[1], round(r[2]), round(r[3],1))
PY
