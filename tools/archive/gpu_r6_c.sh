#!/bin/bash
O=gpurun_out; mkdir -p $O
tools/ubench/lds_probe_diag | tail -3
tools/ubench/urem24_check | tee $O/r6c_urem24.txt
timeout 1500 python -m pytest tests/test_gpu_lds_guard.py -q -m gpu -x > $O/r6c_tests1.txt 2>&1; echo "lds guard rc=$?"; tail -3 $O/r6c_tests1.txt
timeout 900 python bench.py > $O/r6c_bench.json 2> $O/r6c_bench.err; echo "bench rc=$?"; tail -c 600 $O/r6c_bench.err
timeout 1800 python -m pytest tests -q -m gpu -x > $O/r6c_full.txt 2>&1; echo "full suite rc=$?"; tail -5 $O/r6c_full.txt
timeout 900 python tools/gpu_flake_hunt2.py 60 --world 2 --transport fakerccl --jobs 3 --seconds 500 > $O/r6c_flake_fake.txt 2>&1; echo "flake fakerccl rc=$?"; tail -4 $O/r6c_flake_fake.txt
