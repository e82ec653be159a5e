#!/usr/bin/env python3
"""--fasta file -> sketch, the line pump on the device against the host's parser threads.
usage: fa_device_rate.py [contigs (2000)] [length (500000)] [runs (3)]     FA_HOST=0: skip the host-parser runs"""
import hashlib, os, sys, tempfile, time, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import hulk_amd
from hulk_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = tempfile.mkdtemp(dir="/dev/shm")
path = os.path.join(d, "c.fa")
W = 60
short = L <= 1000                                  # reads in FASTA clothing: one sequence line per record, written in bulk
if short:
    with open(path, "wb") as fh:
        for first in range(0, n, 500_000):
            m = min(500_000, n - first)
            bases = synth.reads_numpy(first, m, L)[0][:m * L].reshape(m, L)
            rec = np.empty((m, 9 + L + 1), dtype=np.uint8)
            ids = np.char.zfill(np.arange(first, first + m).astype("U7"), 7)
            rec[:, 0] = ord(">"); rec[:, 1:8] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(m, 7); rec[:, 8] = ord("\n")
            rec[:, 9:9 + L] = bases; rec[:, 9 + L] = ord("\n")
            fh.write(rec.tobytes())
with open(path, "ab" if short else "wb") as fh:
    for i in range(0 if short else n):
        seq = synth.reads_numpy(i, 1, L)[0][:L]
        rows = (L + W - 1) // W
        buf = np.full((rows, W + 1), ord("\n"), dtype=np.uint8)
        pad = np.zeros(rows * W, dtype=np.uint8); pad[:L] = seq
        buf[:, :W] = pad.reshape(rows, W)
        body = buf.tobytes()
        if L % W:
            body = body[:len(body) - (W - L % W) - 1] + b"\n"
        fh.write(b">c%d\n" % i + body)
size = os.path.getsize(path)
print("file: %d contigs x %d = %.1f MB" % (n, L, size / 1e6), flush=True)
for label, flags in (("device", 0), ("host parser", _lib.HULK_INGEST_HOST_PARSER)):
    if flags and os.environ.get("FA_HOST") == "0":
        continue
    for r in range(runs):
        sk = hulk_amd.GpuSketcher(21, 9, 512, interval=0)
        t0 = time.perf_counter(); st = sk.sketch_files([path], fasta=True, opts={"flags": flags, "block_bytes": int(os.environ.get("FA_BLOCK", "0"))}); sk.finish(); dt = time.perf_counter() - t0
        m = hashlib.md5(sk.sketch()[0].astype("<u8").tobytes()).hexdigest()[:8]
        sk.close()
        print("%-12s run %d: %.1f ms, %.3g bases/s, %.2f GB/s of file | %d seqs %d bases %d lines | md5 %s" % (label, r, dt * 1e3, n * L / dt, size / dt / 1e9, st["n_seqs"], st["total_len"], st["n_lines"], m), flush=True)
shutil.rmtree(d)
