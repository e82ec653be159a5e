# per-launch durations of the flush kernels over the first steps of a context (rocprofv3 kernel trace, csv)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_flush; rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/tools/shard_projection.py --worlds 1 --steps 20 > $OUT/out.json 2> $OUT/err.txt
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/trace_flush/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    m = re.search(r'\bk_[a-z0-9_]+', n); n = m.group(0) if m else n
    by[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in ('k_count_used', 'k_flush_decide', 'k_cms_segsum', 'k_cms_base', 'k_cms_freq', 'k_rcp_extrema', 'k_scan_test', 'k_cws_scan', 'k_cws_resolve', 'k_cws_apply', 'k_minimizer_fast', 'k_jump_bin', 'k_nibble_hist'):
    v = by.get(k, [])
    print(k, len(v), ' '.join('%.0f' % x for x in v[:24]))
PY
rm -rf $OUT
