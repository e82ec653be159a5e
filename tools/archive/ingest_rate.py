#!/usr/bin/env python3
"""Host ingest rate: writes a synthetic FASTQ (150 bp reads) under .scratch/ and times
hulk_parse_files (parse only) and, with --gpu, hulk_sketch_files end to end.
usage: ingest_rate.py [n_reads] [--gz | --bgzf] [--gpu] [--threads T]"""
import argparse
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hulk_amd import ingest, synth

ap = argparse.ArgumentParser()
ap.add_argument("n_reads", nargs="?", type=int, default=1_000_000)
ap.add_argument("--gz", action="store_true")
ap.add_argument("--bgzf", action="store_true", help="bgzip container (members of 64 KiB inflated side by side) instead of one gzip member")
ap.add_argument("--gpu", action="store_true")
ap.add_argument("--threads", type=int, default=0)
a = ap.parse_args()
os.makedirs(os.path.join(ROOT, ".scratch"), exist_ok=True)
path = os.path.join(ROOT, ".scratch", "synth_%d.fq%s" % (a.n_reads, ".bgzf.gz" if a.bgzf else ".gz" if a.gz else ""))


class BgzfWriter:
    """bgzip's container, level 1: 65280-byte members with the BC extra subfield, and the empty end-of-file member"""
    def __init__(self, p):
        self.fh, self.buf = open(p, "wb"), bytearray()

    def member(self, piece):
        import zlib
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = c.compress(bytes(piece)) + c.flush()
        self.fh.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\x00BC\x02\x00" + (18 + len(body) + 8 - 1).to_bytes(2, "little") + body +
                      (zlib.crc32(bytes(piece)) & 0xffffffff).to_bytes(4, "little") + len(piece).to_bytes(4, "little"))

    def write(self, data):
        self.buf += data
        while len(self.buf) >= 65280:
            self.member(self.buf[:65280]); del self.buf[:65280]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.buf:
            self.member(self.buf)
        self.member(b"")
        self.fh.close()


if not os.path.exists(path):
    L = 150
    op = BgzfWriter if a.bgzf else (lambda p: gzip.open(p, "wb", compresslevel=1)) if a.gz else (lambda p: open(p, "wb"))
    with op(path) as fh:
        qual = b"I" * L
        for first in range(0, a.n_reads, 100000):
            n = min(100000, a.n_reads - first)
            bases, _ = synth.reads_numpy(first, n, L)
            b = bases[:n * L].tobytes()
            fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (first + i, b[i * L:(i + 1) * L], qual) for i in range(n)))
size = os.path.getsize(path)
for rep in range(2):
    t0 = time.time()
    b, o, st = ingest.parse_files([path], threads=a.threads, collect=False)
    dt = time.time() - t0
print("parse: %d reads, %.1f MB text in %.3f s native (%.2f s with numpy copies) -> %.2e reads/s, %.2f GB/s of text"
      % (st["n_seqs"], st["bytes_in"] / 1e6, st["seconds"], dt, st["n_seqs"] / st["seconds"], st["bytes_in"] / st["seconds"] / 1e9))
if a.gpu:
    import hulk_amd
    g = hulk_amd.GpuSketcher(21, 9, 512, interval=100000)
    g.sketch_files([path], threads=a.threads)      # warm: tables, staging
    g.finish(); g.close()
    g = hulk_amd.GpuSketcher(21, 9, 512, interval=100000)
    t0 = time.time()
    st = g.sketch_files([path], threads=a.threads)
    g.finish()
    dt = time.time() - t0
    print("sketch_files: %d reads in %.3f s (%.3f s inside hulk_sketch_files) -> %.2e reads/s end to end (file %.1f MB on disk)"
          % (st["n_seqs"], dt, st["seconds"], st["n_seqs"] / dt, size / 1e6))
    g.close()
