#!/bin/bash
O=gpurun_out; mkdir -p $O
tools/ubench/lds_probe_diag > $O/r6b_probe_diag.txt 2>&1; tail -40 $O/r6b_probe_diag.txt
tools/ubench/lds_atomic_order 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_lds_guard.py tests/test_gpu_fake_rccl.py tests/test_gpu_smash.py tests/test_gpu_tablegen.py -q -m gpu > $O/r6b_tests1.txt 2>&1; echo "tests1 rc=$?"
tail -30 $O/r6b_tests1.txt
