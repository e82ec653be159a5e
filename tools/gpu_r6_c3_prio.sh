#!/bin/bash
# C3: wave priority of the count-min replay kernels (HULK_CMSD_PRIO), the flush held back behind the next batch's k_minimizer_fast
# (HULK_C3_HOLD), one and two work lanes -> profiles/r06_c3_prio.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
out=gpurun_out/r06_c3_prio.txt; : > $out
export HULK_LIB=exp
run() { echo "== $1" >> $out; shift; env "$@" timeout 200 python tools/c3_probe.py 2>&1 | grep -v amdgpu.ids >> $out; }
run "default (one lane, priority 0)" X=1
run "priority 3" HULK_CMSD_PRIO=3
run "priority 1" HULK_CMSD_PRIO=1
run "hold" HULK_C3_HOLD=1
run "hold + priority 3" HULK_C3_HOLD=1 HULK_CMSD_PRIO=3
run "two lanes" C3_LANES=2
run "two lanes + priority 3" C3_LANES=2 HULK_CMSD_PRIO=3
run "two lanes + hold + priority 3" C3_LANES=2 HULK_C3_HOLD=1 HULK_CMSD_PRIO=3
run "default again" X=1
cat $out
