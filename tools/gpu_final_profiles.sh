# Final-state evidence for profiles/: kernel stats (overlapped, serial, serial without pruning) + PMC json.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final; rm -rf $O; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" bash tools/gpu_prof_bench.sh > $O/$name.txt 2>&1
  python tools/rocprof_summary.py gpurun_out/prof_bench/b_results.db $O/$name.md "$TITLE" "$CMD" > /dev/null
  rm -rf gpurun_out/prof_bench
}
TITLE="Round 1 (e): final kernels, default configuration (two streams, CWS-scan pruning on)" CMD="python bench.py --no-cpu-baseline --single-pass" run overlap A=1
TITLE="Round 1 (e): each kernel alone (HULK_NO_OVERLAP=1), pruning on" CMD="HULK_NO_OVERLAP=1 python bench.py --no-cpu-baseline --single-pass" run serial HULK_NO_OVERLAP=1
TITLE="Round 1 (e): each kernel alone (HULK_NO_OVERLAP=1), pruning off (HULK_NO_PRUNE=1)" CMD="HULK_NO_OVERLAP=1 HULK_NO_PRUNE=1 python bench.py --no-cpu-baseline --single-pass" run serial_noprune HULK_NO_OVERLAP=1 HULK_NO_PRUNE=1
bash tools/gpu_pmc.sh > $O/pmc_print.txt 2>&1
python tools/pmc_to_json.py gpurun_out/pmc $O/r01_pmc.json
rm -rf gpurun_out/pmc
python bench.py > $O/bench_final.json 2> $O/bench_final.err
ls -la $O
