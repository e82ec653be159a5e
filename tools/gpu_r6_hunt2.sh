#!/bin/bash
# round 6: the rest of the hunt (VERDICT r5 asked for N = 300 at world two, N = 100 at world eight)
O=gpurun_out; mkdir -p $O
timeout 1500 python tools/gpu_flake_hunt2.py 160 --world 2 --transport gloo --jobs 3 --seconds 900 > $O/r6i_flake_gloo.txt 2>&1; tail -2 $O/r6i_flake_gloo.txt
timeout 1200 python tools/gpu_flake_hunt2.py 30 --world 8 --transport gloo --jobs 1 --seconds 420 > $O/r6i_flake_w8_gloo.txt 2>&1; tail -2 $O/r6i_flake_w8_gloo.txt
timeout 1200 python tools/gpu_flake_hunt2.py 30 --world 8 --transport fakerccl --jobs 1 --seconds 420 > $O/r6i_flake_w8_fake.txt 2>&1; tail -2 $O/r6i_flake_w8_fake.txt
