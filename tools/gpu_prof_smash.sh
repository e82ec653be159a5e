# BASELINE C5 (hulk smash: 1024 sketches x sketchSize 2048): rocprofv3 kernel stats of tools/smash_c5.py -> gpurun_out/smash/
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/smash; rm -rf $O; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python $GRAFT_REPO_ROOT/tools/smash_c5.py > $O/out.txt 2> $O/err.txt
cd $GRAFT_REPO_ROOT; cat $O/out.txt
python tools/rocprof_summary.py $(ls $O/prof/*/*results.db $O/prof/*results.db 2>/dev/null | head -1) $O/${R:-r05}_c5_smash_kernel_stats.md "Round ${R#r0}: BASELINE C5 — hulk smash, 1024 sketches x sketchSize 2048, weighted Jaccard and Jaccard" "python tools/smash_c5.py" | grep k_smash
rm -rf $O/prof
