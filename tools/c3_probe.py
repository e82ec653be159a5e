#!/usr/bin/env python3
"""C3 (k = 31, sketchSize 1024, decay 0.02, interval 100k) on an HBM-resident sample, short form of bench.py's c3 leg:
overlapped reads/s, then every kernel alone (hulk_set_profiling(32)).  usage: c3_probe.py [reads (default 16 M)]
Prints one line per mode; the sketch's md5 so that variants can be compared."""
import hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hulk_amd
from hulk_amd import _lib, synth
READS = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
LANES = int(os.environ.get("C3_LANES", "0"))
k3, w3, S3, I3, BATCH, L = 31, 9, 1024, 100_000, 16, 150
stp = I3 * BATCH
bufs = [synth.reads_torch(s * stp, stp, L, device="cuda:0") for s in range(4)]
torch.cuda.synchronize()
for serial in (False, True):
    sk = hulk_amd.GpuSketcher(k3, w3, S3, interval=I3, decay_ratio=0.02, device=0, batch=BATCH, work_lanes=LANES, flags=_lib.HULK_FLAG_NO_OVERLAP if serial else 0)
    b, o = bufs[0]
    sk.add_reads_device(b.data_ptr(), o.data_ptr(), stp, L, b.numel()); sk.synchronize()
    if serial:
        sk.set_profiling(32)
    t0 = time.perf_counter(); done, i = 0, 1
    while done < READS:
        b, o = bufs[i % 4]
        sk.add_reads_device(b.data_ptr(), o.data_ptr(), stp, L, b.numel()); done += stp; i += 1
    sk.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    nb = done / stp
    line = "%s: %.4g reads/s, %.3f ms per batch" % ("alone" if serial else "overlapped", done / ms * 1e3, ms / nb)
    if serial:
        tbl = sk.profile_table(); sk.set_profiling(0)
        line += " | " + ", ".join("%s %.0f" % (kk, v[1] / nb * 1e3) for kk, v in sorted(tbl.items(), key=lambda kv: -kv[1][1])[:7])
    sk.finish()
    mins, _ = sk.sketch()
    print(line, "| md5", hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()[:8], flush=True)
    sk.close(); torch.cuda.empty_cache()
