"""PCIe-inclusive rate: reads handed over as HOST buffers through hulk_add_reads (pageable numpy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hulk_amd
from hulk_amd import synth
n = 1_000_000
sk = hulk_amd.GpuSketcher(21, 9, 512, interval=100_000)
batches = [synth.reads_numpy(i * n, n, 150) for i in range(3)]
sk.add_reads(*batches[0]); sk.counters()
t0 = time.perf_counter()
for b in batches[1:]:
    sk.add_reads(*b)
sk.counters()
dt = time.perf_counter() - t0
print(f"host-buffer path: {2 * n / dt:.3e} reads/s ({dt * 1e3 / 2:.1f} ms per 1e6 reads, {2 * n * 158 / dt / 1e9:.1f} GB/s of input)")
sk.finish(); sk.close()
