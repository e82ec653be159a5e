cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=/tmp/tl; rm -rf $OUT
cd /tmp && rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hulk' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.search(r'(k_\w+)', r['Kernel_Name']).group(1), r.get('Queue_Id', '?')))
rows.sort()
# print the timeline of one steady-state step window
fast = [r for r in rows if r[2] == "k_minimizer_fast"]; t0 = fast[-3][0]
for s, e, n, q in rows:
    if t0 <= s < t0 + 3_200_000:
        print(f"{(s - t0) / 1000:9.1f} {(e - t0) / 1000:9.1f}  {(e - s) / 1000:8.1f} us  q{q}  {n}")
PY
