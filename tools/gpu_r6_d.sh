#!/bin/bash
# round 6, call D: C3 scheduling experiments, the new bench legs, the long-sequence test, urem24, more of the hunt
O=gpurun_out; mkdir -p $O
tools/ubench/urem24_check | tee $O/r6d_urem24.txt
c3() { python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g' % (d['reads_per_s']/1e9), end=' ')"; }
export HULK_LIB=exp
{
echo "# C3-shaped rate (tools/run_config.py: k=31, S=1024, decay 0.02, 32 M reads, 16 intervals per batch), profiling build, 1e9 reads/s, three runs each"
echo "default (one work lane, all CUs):                     $(c3; c3; c3)"
for n in 16 32 48 96; do echo "binning lanes barred from $n of the 256 CUs (HULK_K1_CU_FREE=$n): $(HULK_K1_CU_FREE=$n c3; HULK_K1_CU_FREE=$n c3; HULK_K1_CU_FREE=$n c3)"; done
echo "k_minimizer_fast gated on the 112 KB kernels of the flush two batches back (HULK_C3_GATE=1): $(HULK_C3_GATE=1 c3; HULK_C3_GATE=1 c3; HULK_C3_GATE=1 c3)"
echo "gate + 16 CUs free:                                   $(HULK_C3_GATE=1 HULK_K1_CU_FREE=16 c3; HULK_C3_GATE=1 HULK_K1_CU_FREE=16 c3)"
echo "two lanes:                                            $(c3 --lanes 2; c3 --lanes 2)"
echo "two lanes + gate:                                     $(HULK_C3_GATE=1 c3 --lanes 2; HULK_C3_GATE=1 c3 --lanes 2)"
} > $O/r6d_c3_sched.txt 2>&1
cat $O/r6d_c3_sched.txt
unset HULK_LIB
timeout 900 python -m pytest tests/test_gpu_fullsize.py::test_long_sequences_against_oracle_and_split_invariance -x -q -m gpu > $O/r6d_longtest.txt 2>&1; echo "long test rc=$?"; tail -3 $O/r6d_longtest.txt
timeout 900 python bench.py --no-c3 --no-c5 --no-long-reads --no-e2e --no-cpu-baseline > $O/r6d_bench.json 2> $O/r6d_bench.err; echo "bench rc=$?"; tail -c 300 $O/r6d_bench.err
timeout 1200 python tools/gpu_flake_hunt2.py 90 --world 2 --transport gloo --jobs 3 --seconds 480 > $O/r6d_flake_gloo.txt 2>&1; echo "flake gloo rc=$?"; tail -3 $O/r6d_flake_gloo.txt
