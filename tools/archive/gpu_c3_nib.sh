#!/bin/bash
# C3-shaped rate against the 4-bit histogram's range size / workgroup size / keys per part (profiling build switches)
export HULK_LIB=exp
c3() { python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 24000000 --interval 100000 --batch 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g' % (d['reads_per_s']/1e9), end=' ')"; }
echo "defaults: $(c3; c3)"
for rl in 14 15 16 17 18; do for bl in 256 1024; do echo "HULK_NIB_RLOG=$rl HULK_NIB_BLOCK=$bl: $(HULK_NIB_RLOG=$rl HULK_NIB_BLOCK=$bl c3)"; done; done
for ky in 65536 262144 524288; do echo "HULK_NIB_KEYS=$ky: $(HULK_NIB_KEYS=$ky c3)"; done
echo "HULK_NO_NIBBLE=1 (exact range histogram): $(HULK_NO_NIBBLE=1 c3)"
