# C3-shaped configuration (k=31, sketchSize=1024, decay 0.02): rate with the two streams, then rocprofv3 kernel stats with every kernel alone
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c3; mkdir -p $O
python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 > $O/rate.json 2> $O/rate.err; cat $O/rate.json
python tools/run_config.py --no-prune --k 31 --S 1024 --decay 0.02 --reads 8000000 --interval 100000 --batch 16 > $O/rate_noprune.json 2>> $O/rate.err; cat $O/rate_noprune.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg; rm -rf $OUT; mkdir -p $OUT
# 32 timed batches + the warm-up batch; the steady-state table leaves out everything in front of the second k_minimizer_fast launch
cd /tmp && HULK_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o b -- python $GRAFT_REPO_ROOT/tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 52800000 --interval 100000 --batch 16 > $OUT/out.json 2> $OUT/err.txt
cd $GRAFT_REPO_ROOT
python tools/rocprof_steady.py $(ls $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv 2>/dev/null | head -1) $O/${R:-r06}_c3_kernel_stats_serial.md "Round ${R#r0}: C3-shaped configuration (k=31, sketchSize=1024, decay 0.02, 22.7 GB of CWS tables), each kernel alone, steady state" "HULK_NO_OVERLAP=1 python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 52800000 --interval 100000 --batch 16" | grep -E "^\| k_|^Sum" | head -40
rm -rf $OUT
