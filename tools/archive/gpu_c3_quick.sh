#!/bin/bash
# C3 quick look: decay parity tests, the C3-shaped rate (twice), per-kernel times of the decay flush with every kernel alone
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "drift or decay or k31 or c3" > $O/gpu_tests_c3q.txt 2>&1; echo "rc=$?" >> $O/gpu_tests_c3q.txt; tail -3 $O/gpu_tests_c3q.txt | cut -c1-200
c3() { python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 32000000 --interval 100000 --batch 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e reads/s (%.3f ms per batch)' % (d['reads_per_s'], d['ms']/(d['reads_timed']/1.6e6)))"; }
echo "c3: $(c3)"; echo "c3 again: $(c3)"
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, hulk_amd
from hulk_amd import synth, _lib
sk = hulk_amd.GpuSketcher(31, 9, 1024, interval=100000, decay_ratio=0.02, batch=16, flags=_lib.HULK_FLAG_NO_OVERLAP)
step = 1600000
bufs = [synth.reads_torch(s * step, step, 150) for s in range(3)]
torch.cuda.synchronize()
for i in range(3): sk.add_reads_device(bufs[i][0].data_ptr(), bufs[i][1].data_ptr(), step, 150, bufs[i][0].numel())
sk.synchronize(); sk.set_profiling(True)
for i in range(12): b, o = bufs[i % 3]; sk.add_reads_device(b.data_ptr(), o.data_ptr(), step, 150, b.numel())
sk.synchronize()
for k in ("k_cmsd_freq", "k_minimizer_fast", "k_jump_bin", "k_cws_scan"):
    n, ms = sk.get_profile(k); print(k, n, "%.1f us" % (ms / max(n, 1) * 1e3))
PY
