cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 0 1 4 8 24; do
OUT=/tmp/k1s_$d; rm -rf $OUT
cd /tmp && HULK_K1_DEBUG=$d rocprofv3 --kernel-trace --stats -d $OUT -o k -- python $GRAFT_REPO_ROOT/tools/k1_ablate.py child > /dev/null 2>&1
python - <<PY
import sqlite3,glob
for f in glob.glob('$OUT/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    print('dbg=$d', [(r[0].split('(')[0][-22:], round(r[2],1)) for r in c.execute("select name,total_calls,average from top_kernels") if 'k_minimizer_fast' in r[0] or 'k_jump' in r[0] or 'k_range' in r[0]])
PY
done
