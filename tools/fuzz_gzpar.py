#!/usr/bin/env python3
"""Randomised differential test of the parallel single-member gzip reader (GzPar, hulk_ingest.hip / par_inflate.h) against
the one-thread reader (HULK_GZ_PAR=0), which tests/test_ingest_cpu.py and tools/fuzz_ingest.py hold against the restated
reference: FASTQ-like text of 0.2-6 MB deflated at random levels / strategies / memLevels (block sizes), as one member,
several members (the parallel reader ends a member itself and goes on with the next), with stored and fixed blocks, sync-flush points (empty stored blocks, as pigz writes them), trailing bytes,
truncations and flipped bits — both readers must deliver the same reads or the same message.  Chunks of 32-128 KiB
(HULK_GZ_PAR_CHUNK; sometimes 8 KiB, less than a block: the chain keeps breaking and the reader gives up) so that every file is
many chunks and several batches.  CPU only.
usage: fuzz_gzpar.py [n_cases] [seed]      (child mode: fuzz_gzpar.py --child file...)"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from hulk_amd import ingest
    from hulk_amd._lib import HulkError
    for p in sys.argv[2:]:
        try:
            b, o, st = ingest.parse_files([p])
            print(json.dumps([p, int(st["n_seqs"]), hashlib.md5(b.tobytes()).hexdigest(), hashlib.md5(o.tobytes()).hexdigest()]), flush=True)
        except HulkError as e:
            print(json.dumps([p, "error", e.message]), flush=True)
    sys.exit(0)

import numpy as np

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def fastq(n_reads):
    acgt = np.frombuffer(b"ACGTN", np.uint8)
    qual = np.frombuffer(b"FFFFFFF:,F#", np.uint8)
    L = int(rng.choice([36, 100, 150, 251]))
    low_complexity = rng.random() < 0.2                          # long matches, a high ratio (the symbol buffer can run out)
    out = []
    for i in range(n_reads):
        s = (acgt[rng.integers(0, 4, 4)].tobytes() * (L // 4 + 1))[:L] if low_complexity else acgt[rng.integers(0, 5 if rng.random() < 0.1 else 4, L)].tobytes()
        out.append(b"@m%d:%d/1\n%s\n+\n%s\n" % (seed, i, s, qual[rng.integers(0, 11, L)].tobytes() if not low_complexity else b"F" * L))
    return b"".join(out)


def deflate(data):
    level = int(rng.choice([0, 1, 1, 3, 6, 6, 9]))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    c = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.integers(1, 10)), strategy)
    out, at = [], 0
    flushy = rng.random() < 0.3
    while at < len(data):
        n = int(rng.integers(1, 1 << 18)) if flushy else len(data)
        out.append(c.compress(data[at:at + n])); at += n
        if flushy and at < len(data):
            out.append(c.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK if hasattr(zlib, "Z_BLOCK") else zlib.Z_SYNC_FLUSH]))))
    out.append(c.flush())
    return b"".join(out)


def member(data):
    flags = int(rng.choice([0, 0, 8, 4 | 8 | 16]))
    hdr = bytearray(b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\0\xff")
    if flags & 4: hdr += b"\x05\x00hello"
    if flags & 8: hdr += b"reads.fq\0"
    if flags & 16: hdr += b"a comment\0"
    return bytes(hdr) + deflate(data) + (zlib.crc32(data) & 0xffffffff).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")


def run(paths, env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + paths, capture_output=True, text=True, env=e, timeout=3600)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("[")]
    if len(rows) != len(paths):
        print("child failed:", r.stderr[-2000:])
        sys.exit(2)
    import re
    stats = [tuple(map(int, m)) for m in re.findall(r"(\d+) batches, (\d+) chunks counted / (\d+) decoded, \d+ bytes of text, (\d+) members ended", r.stderr)]
    return {row[0]: row[1:] for row in rows}, stats


bad = 0
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    paths, kinds = [], {}
    for case in range(n_cases):
        text = fastq(int(rng.choice([2000, 6000, 20000, 40000])))
        r = rng.random()
        if r < 0.5:
            blob, kind = member(text), "one member"
        elif r < 0.65:                                               # 2-4 members (cut anywhere, an empty one now and then), trailing bytes
            cuts = sorted(int(x) for x in rng.integers(0, len(text) + 1, int(rng.integers(1, 4))))
            parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
            if rng.random() < 0.3:
                parts.insert(int(rng.integers(0, len(parts) + 1)), b"")
            blob, kind = b"".join(member(x) for x in parts) + (b"\0\0junk" if rng.random() < 0.5 else b""), "%d members" % len(parts)
        elif r < 0.8:
            blob = bytearray(member(text)); at = int(rng.integers(20, len(blob))); blob[at] ^= 1 << int(rng.integers(0, 8))
            blob, kind = bytes(blob), "flipped bit at %d" % at
        elif r < 0.92:
            blob = member(text); blob, kind = blob[:int(rng.integers(1, len(blob)))], "truncated"
        else:
            raw = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()       # not text: no candidate anywhere
            blob, kind = member(text[:len(text) // 2] + raw + text[len(text) // 2:]), "binary in the middle"
        p = os.path.join(td, "c%d.fq.gz" % case)
        open(p, "wb").write(blob)
        paths.append(p); kinds[p] = kind
    chunk = str(int(rng.choice([8192, 32768, 65536, 65536, 131072])))
    threads = str(int(rng.choice([2, 3, 8])))
    want, _ = run(paths, {"HULK_GZ_PAR": "0"})
    got, stats = run(paths, {"HULK_GZ_PAR_CHUNK": chunk, "HULK_GZ_THREADS": threads, "HULK_INGEST_TRACE": "1"})
    for p in paths:
        if want[p] != got[p]:
            bad += 1
            print("MISMATCH", os.path.basename(p), kinds[p], "chunk", chunk, "threads", threads, "::", want[p], "|", got[p], flush=True)
    n_err = sum(1 for p in paths if want[p][0] == "error")
print(f"{n_cases} cases (seed {seed}, chunk {chunk}, threads {threads}; {n_err} of them end in a message; the parallel reader ran on {len(stats)} files, "
      f"{sum(s[0] for s in stats)} batches, {sum(s[1] for s in stats)} chunks counted of {sum(s[2] for s in stats)} decoded, {sum(s[3] for s in stats)} members ended in it), {bad} mismatches")
sys.exit(1 if bad else 0)
