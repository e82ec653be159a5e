#!/usr/bin/env python3
"""FASTQ file -> sketch (C2 parameters), the line machine on the device; HULK_INGEST_TRACE=1 prints where the calling thread waits.
usage: fq_device_rate.py [reads (8000000)] [runs (4)]"""
import hashlib, os, sys, tempfile, time, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import hulk_amd
from hulk_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 150
d = tempfile.mkdtemp(dir="/dev/shm")
path = os.path.join(d, "r.fq")
with open(path, "wb") as fh:
    for first in range(0, n, 500_000):
        m = min(500_000, n - first)
        bases = synth.reads_numpy(first, m, L)[0][:m * L].reshape(m, L)
        rec = np.empty((m, 8 + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
        ids = np.char.zfill(np.arange(first, first + m).astype("U7"), 7)
        rec[:, 0] = ord("@"); rec[:, 1:8] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(m, 7); rec[:, 8] = ord("\n")
        rec[:, 9:9 + L] = bases; rec[:, 9 + L] = ord("\n"); rec[:, 10 + L] = ord("+"); rec[:, 11 + L] = ord("\n")
        rec[:, 12 + L:12 + 2 * L] = ord("I"); rec[:, 12 + 2 * L] = ord("\n")
        fh.write(rec.tobytes())
size = os.path.getsize(path)
print("file: %d reads, %.1f MB" % (n, size / 1e6), flush=True)
for r in range(runs):
    sk = hulk_amd.GpuSketcher(21, 9, 512, interval=100_000)
    opts = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("FQ_OPTS", "").split(",") if kv)}
    t0 = time.perf_counter(); st = sk.sketch_files([path], opts=opts or None); sk.finish(); dt = time.perf_counter() - t0
    m5 = hashlib.md5(sk.sketch()[0].astype("<u8").tobytes()).hexdigest()[:8]
    sk.close()
    print("run %d: %.1f ms, %.3g reads/s, %.1f GB/s of file | md5 %s" % (r, dt * 1e3, n / dt, size / dt / 1e9, m5), flush=True)
shutil.rmtree(d)
