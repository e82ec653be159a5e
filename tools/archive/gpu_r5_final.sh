#!/bin/bash
# round 5: the whole GPU suite, smoke, the profiles of the round (kernel stats overlapped / serial / unpruned, PMC, C3, C5), a fuzz soak
O=gpurun_out; mkdir -p $O
export R=r05
timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_final_profiles.sh > $O/final_profiles.log 2>&1; ls gpurun_out/final_r05 | head -20
bash tools/gpu_prof_c3.sh 2>&1 | grep -E "^\| k_(cmsd|cws_scan|minimizer_fast|jump_bin)" | head
bash tools/gpu_prof_smash.sh 2>&1 | tail -3
bash tools/gpu_soak.sh 9700 2 | tail -30
FUZZ_SECONDS=100 timeout 200 python tools/fuzz_devparse.py 100000 9710 2>&1 | tail -1 | tee -a gpurun_out/soak/soak_9700.txt
