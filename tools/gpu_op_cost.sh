# VALU issue cost by two methods (tools/ubench/op_cost.hip: HIP events / nominal clock; op_cost2.hip: in-kernel s_memtime per wave,
# grouped by the SIMD the wave ran on — cycles, no clock assumed) + the clock another way: GRBM_GUI_ACTIVE per kernel of the same binary (rocprofv3 --pmc,
# its own pass) over the kernel's duration from the kernel trace.   gpurun -- 'bash tools/gpu_op_cost.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
make -C tools/ubench op_cost op_cost2 > /dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/op_cost; rm -rf $O; mkdir -p $O
{
echo "== method 1: whole launches by HIP events, cycles at a NOMINAL 2.4 GHz (tools/ubench/op_cost.hip)"
tools/ubench/op_cost
echo
echo "== method 2: every wave times its own block with s_memtime (shader-clock ticks) and reports its SIMD (tools/ubench/op_cost2.hip)"
tools/ubench/op_cost2
} > $O/op_cost.txt 2>&1
cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/ubench/op_cost2 > /dev/null 2> $O/pmc.err
cd $GRAFT_REPO_ROOT
python - <<'PY' >> $O/op_cost.txt
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "op_cost")
cnt = glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True)
trc = glob.glob(O + "/pmc/**/*kernel_trace.csv", recursive=True)
print()
print("== clock another way: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, rocprofv3 --pmc pass over op_cost2")
if not cnt or not trc:
    print("no counter output", cnt, trc)
else:
    dur = {}
    for r in csv.DictReader(open(trc[0])):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(cnt[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        name, ns = dur.get(r["Dispatch_Id"], (r.get("Kernel_Name", "?"), 0))
        if ns > 0:
            per[name.split("(")[0]].append(float(r["Counter_Value"]) / 8.0 / ns)
    for name, v in per.items():
        v.sort()
        print(f"{name:<22s} {len(v):3d} launches  median {v[len(v)//2]:.3f} GHz  (min {v[0]:.3f}, max {v[-1]:.3f})")
PY
rm -rf $O/pmc
cat $O/op_cost.txt
