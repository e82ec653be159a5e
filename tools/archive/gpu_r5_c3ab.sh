#!/bin/bash
# round 5, call 6: C3 A/B of the count-min replay's group size (profiling build), C3 kernel stats, two contexts in one process
O=gpurun_out; mkdir -p $O
c3() { python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e reads/s (%.3f ms per batch)' % (d['reads_per_s'], d['ms']/(d['reads_timed']/1.6e6)))"; }
echo "c3 shipping library:            $(c3)"
echo "c3 profiling build, defaults:   $(HULK_LIB=exp c3)"
echo "c3 profiling build, FG16:       $(HULK_LIB=exp HULK_CMSD_FG16=1 c3)"
echo "c3 profiling build, FG16 again: $(HULK_LIB=exp HULK_CMSD_FG16=1 c3)"
echo "c3 profiling build, chain form: $(HULK_LIB=exp HULK_CMSD_CHAIN=1 c3)"
echo "c3 profiling build, grid scan:  $(HULK_LIB=exp HULK_SCAN_GRID=1 c3)"
echo "c3 profiling build, jump lds 16K: $(HULK_LIB=exp HULK_JUMP_LDS=16384 c3)"
R=r05 bash tools/gpu_prof_c3.sh 2>&1 | grep -E "^\| k_(cmsd|cms_|cws_scan|minimizer_fast|jump|nibble|elem|slot|scan|rcp|cws_resolve)" | head -24
python tools/two_ctx_idle.py 2>&1 | grep -v amdgpu.ids | tee $O/r05_two_ctx.txt
