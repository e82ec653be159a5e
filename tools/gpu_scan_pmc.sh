cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export HULK_NO_OVERLAP=1
OUT=/tmp/sp; rm -rf $OUT
cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o f --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections,re
for f in glob.glob('/tmp/sp/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(float); n=collections.Counter()
    for row in csv.DictReader(open(f)):
        m=re.search(r'(k_\w+)',row['Kernel_Name'])
        if m and 'hulk' in row['Kernel_Name']: agg[m.group(1)]+=float(row['Counter_Value']); n[m.group(1)]+=1
    for k in agg: print(k, round(agg[k]/n[k]*2*1024/1e6,1), 'MB fetched per launch (corrected x2)')
for f in glob.glob('/tmp/sp/**/*kernel_trace.csv', recursive=True):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_\w+)',r['Kernel_Name'])
        if m and 'hulk' in r['Kernel_Name']: d[m.group(1)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
    for k,v in d.items(): print(k, 'avg us', round(sum(v)/len(v),1))
PY
