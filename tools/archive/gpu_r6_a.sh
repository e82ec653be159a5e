#!/bin/bash
# round 6, call A: the new tests (LDS guard, RCCL branch over the test double, native smash, tablegen comparator, inject in the
# profiling build, world-two contract over both transports), then a first slice of the flake hunt
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lds_guard.py tests/test_gpu_fake_rccl.py tests/test_gpu_smash.py tests/test_gpu_tablegen.py -x -q -m gpu > $O/r6a_tests1.txt 2>&1; echo "tests1 rc=$?" | tee -a $O/r6a_summary.txt
tail -5 $O/r6a_tests1.txt
timeout 1500 python -m pytest "tests/test_gpu_two_rank.py::test_void_or_late_header_block_is_never_a_silent_wrong_sketch" "tests/test_gpu_bench_contract.py::test_bench_world_two_end_to_end_on_one_gpu" -x -q -m gpu > $O/r6a_tests2.txt 2>&1; echo "tests2 rc=$?" | tee -a $O/r6a_summary.txt
tail -5 $O/r6a_tests2.txt
timeout 900 python tools/gpu_flake_hunt2.py 60 --world 2 --transport gloo --jobs 3 --seconds 600 > $O/r6a_flake_gloo.txt 2>&1; echo "flake gloo rc=$?" | tee -a $O/r6a_summary.txt
tail -4 $O/r6a_flake_gloo.txt
