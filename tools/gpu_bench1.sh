set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 100 --warmup 5 > gpurun_out/bench_r01_a.json 2> gpurun_out/bench_r01_a.err; tail -3 gpurun_out/bench_r01_a.err; cat gpurun_out/bench_r01_a.json
python bench.py --steps 20 --warmup 2 --force-collective --no-cpu-baseline 2>&1 | tail -2
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_a -o r01a -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r01_a | head -20
