cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/k1prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/a -o k1 -- python $GRAFT_REPO_ROOT/tools/k1_ablate.py child 2>&1 | tail -5
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/b -o k1b --output-format csv -- python $GRAFT_REPO_ROOT/tools/k1_ablate.py child 2>&1 | tail -5
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA -d $OUT/c -o k1c --output-format csv -- python $GRAFT_REPO_ROOT/tools/k1_ablate.py child 2>&1 | tail -5
find $OUT -type f | head -20
python - <<'PY'
import sqlite3,glob,csv,collections,os
OUT=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/k1prof'
for f in glob.glob(OUT+'/a/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select name,total_calls,average from top_kernels"):
        if 'hulk' in r[0] or 'rocclr' in r[0]: print(r[0][:70], r[1], round(r[2],1))
for sub in ('b','c'):
  for f in glob.glob(OUT+'/%s/**/*counter_collection.csv'%sub, recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']]+=float(row['Counter_Value']); n[(k,row['Counter_Name'])]+=1
    for k,v in agg.items():
        if 'hulk' in k: print(k, {a:round(b/max(n[(k,a)],1)) for a,b in v.items()})
PY
