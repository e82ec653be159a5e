#!/bin/bash
# round 6, call E: the profiles/ evidence of the current state (kernel stats overlapped / serial / unpruned, PMC json, bench line),
# the steady-state C3 table, the hunt at world 8
export R=r06
bash tools/gpu_final_profiles.sh > gpurun_out/r6e_final.txt 2>&1; tail -5 gpurun_out/r6e_final.txt
R=r06 bash tools/gpu_prof_c3.sh > gpurun_out/r6e_c3.txt 2>&1; tail -30 gpurun_out/r6e_c3.txt
timeout 1500 python tools/gpu_flake_hunt2.py 24 --world 8 --transport gloo --jobs 1 --seconds 600 > gpurun_out/r6e_flake_w8_gloo.txt 2>&1; tail -3 gpurun_out/r6e_flake_w8_gloo.txt
timeout 1500 python tools/gpu_flake_hunt2.py 24 --world 8 --transport fakerccl --jobs 1 --seconds 600 > gpurun_out/r6e_flake_w8_fake.txt 2>&1; tail -3 gpurun_out/r6e_flake_w8_fake.txt
