#!/bin/bash
# PMC passes over the C3-shaped configuration, every kernel alone (separate passes, --kernel-trace only): what bounds the decay flush?
export HULK_LIB=${HULK_LIB:-exp}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c3; rm -rf $OUT; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 6400000 --interval 100000 --batch 16 --serial"
cd /tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- $CMD > /dev/null 2> $OUT/$name.err; }
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_INSTS_SMEM
pass valu SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python tools/pmc_to_json.py gpurun_out/pmc_c3 gpurun_out/r05_pmc_c3.json 1600000 > /dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_pmc_c3.json'))
for k in ('k_cmsd_freq','k_cmsd_segsum','k_elem_index','k_nibble_hist','k_nibble_merge','k_minimizer_fast','k_jump_bin'):
    v=d.get(k)
    if v: print(k, {x:v.get(x) for x in ('avg_us','SQ_WAVES','SQ_INSTS_VALU','SQ_INSTS_LDS','SQ_ACTIVE_INST_LDS','SQ_WAIT_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_LDS_ADDR_CONFLICT','SQ_INST_CYCLES_VALU','SQ_BUSY_CYCLES','SQ_WAIT_INST_ANY','SQ_WAVE_CYCLES','SQ_INSTS_VMEM_RD','hbm_bytes_per_launch')})
PY
