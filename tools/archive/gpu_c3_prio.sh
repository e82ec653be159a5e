#!/bin/bash
# C3-shaped rate by work lanes x priority of the flush stream (profiling build: HULK_FLUSH_PRIORITY), three runs each
c3() { python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g' % (d['reads_per_s']/1e9), end=' ')"; }
export HULK_LIB=exp
echo "lanes 1 flush priority lowest (round 4's default with decay): $(c3 --lanes 1; c3 --lanes 1; c3 --lanes 1)"
echo "lanes 2 flush priority lowest:  $(c3 --lanes 2; c3 --lanes 2; c3 --lanes 2)"
echo "lanes 2 flush priority normal:  $(HULK_FLUSH_PRIORITY=0 c3 --lanes 2; HULK_FLUSH_PRIORITY=0 c3 --lanes 2; HULK_FLUSH_PRIORITY=0 c3 --lanes 2)"
echo "lanes 2 flush priority highest: $(HULK_FLUSH_PRIORITY=-1 c3 --lanes 2; HULK_FLUSH_PRIORITY=-1 c3 --lanes 2; HULK_FLUSH_PRIORITY=-1 c3 --lanes 2)"
echo "lanes 1 flush priority highest: $(HULK_FLUSH_PRIORITY=-1 c3 --lanes 1; HULK_FLUSH_PRIORITY=-1 c3 --lanes 1; HULK_FLUSH_PRIORITY=-1 c3 --lanes 1)"
