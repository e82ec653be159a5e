cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export HULK_NO_OVERLAP=1
OUT=/tmp/c3; rm -rf $OUT
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o c -- python $GRAFT_REPO_ROOT/tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 4000000 > /tmp/c3.json 2>/dev/null
cat /tmp/c3.json
python - <<'PY'
import sqlite3,glob
for f in glob.glob('/tmp/c3/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select name,total_calls,total_duration,average from top_kernels"):
        if 'hulk' in r[0] and r[1] < 20: print(r[0][:80], r[1], round(r[2]), round(r[3],1))
PY
