# one rank's share of an 8-rank strong-scaling step (tools/shard_projection.py), each kernel alone: rocprofv3 kernel stats
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/shard; mkdir -p $O
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_shard; rm -rf $OUT; mkdir -p $OUT
cd /tmp && HULK_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d $OUT -o b -- python $GRAFT_REPO_ROOT/tools/shard_projection.py --worlds ${WORLDS:-8} --steps 40 --mode ${MODE:-sharded} > $OUT/out.json 2> $OUT/err.txt
cd $GRAFT_REPO_ROOT; cat $OUT/out.json
python tools/rocprof_summary.py $(ls $OUT/*/*results.db $OUT/*results.db 2>/dev/null | head -1) $O/${R:-r03}_shard8_kernel_stats_serial.md "Round 3: rank 0's share of an 8-rank step of hulk_step_sharded (1.6 M reads of its own, 64 slots, count-min upkeep of 128 intervals; loopback), each kernel alone; mode = ${MODE:-sharded}" "HULK_NO_OVERLAP=1 python tools/shard_projection.py --worlds 8 --steps 40 --mode ${MODE:-sharded}" | grep -E "^\| k_" | head -40
rm -rf $OUT
