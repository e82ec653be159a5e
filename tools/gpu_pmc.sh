# PMC passes for the bench (separate runs, --kernel-trace only — never combined with other trace domains):
export HULK_LIB=${HULK_LIB:-exp}    # the profiling build: HULK_NO_OVERLAP and the other experiment switches exist only there (make EXPERIMENTS=1)
# HBM traffic, VALU activity, LDS activity / bank conflicts, L2 hits.  Results -> gpurun_out/pmc/<pass>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cold --no-e2e --no-c3 --no-c5 --no-long-reads --single-pass --steps 5 --warmup 1 $BENCH_ARGS"
cd /tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- $BENCH > /dev/null 2> $OUT/$name.err; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass valu SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_WAIT_ANY
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_INSTS_SMEM
pass tcc TCC_HIT_sum TCC_MISS_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls $OUT
