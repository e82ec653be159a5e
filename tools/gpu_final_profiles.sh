# Evidence for profiles/ of the current state (round label R, default r04): kernel stats (two streams; every kernel
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
# alone; every kernel alone with the CWS bounds really off), PMC json, bench line.   gpurun -- 'bash tools/gpu_final_profiles.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${R:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$R; rm -rf $O; mkdir -p $O
run() { # name, bench args, env...
  name=$1; args=$2; shift; shift
  env "$@" BENCH_ARGS="$args" bash tools/gpu_prof_bench.sh > $O/$name.txt 2>&1
  python tools/rocprof_summary.py gpurun_out/prof_bench/b_results.db $O/$name.md "$TITLE" "$CMD" > /dev/null
  cp gpurun_out/prof_bench/bench.json $O/$name.bench.json
  rm -rf gpurun_out/prof_bench
}
TITLE="Round ${R#r0}: default configuration (two streams, CWS-scan bounds on)" CMD="python bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --single-pass" run ${R}_kernel_stats "" A=1
TITLE="Round ${R#r0}: each kernel alone (HULK_NO_OVERLAP=1), CWS-scan bounds on" CMD="HULK_NO_OVERLAP=1 python bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --single-pass" run ${R}_kernel_stats_serial "" HULK_NO_OVERLAP=1
TITLE="Round ${R#r0}: each kernel alone (HULK_NO_OVERLAP=1), CWS-scan bounds OFF (bench.py --no-prune = HULK_FLAG_NO_PRUNE on the timed context)" CMD="HULK_NO_OVERLAP=1 python bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --no-prune" run ${R}_kernel_stats_serial_noprune "--no-prune" HULK_NO_OVERLAP=1
HULK_NO_OVERLAP=1 bash tools/gpu_pmc.sh > $O/pmc_print.txt 2>&1
python tools/pmc_to_json.py gpurun_out/pmc $O/${R}_pmc.json
rm -rf gpurun_out/pmc
python bench.py > $O/bench_$R.json 2> $O/bench_$R.err
ls -la $O
