#!/usr/bin/env python3
"""Randomised differential test of the native line pump (hulk_parse_files) against the literal
restatement of the reference (oracle/linepump.py): random line soups, CR/LF mixes, missing final
newlines, empty lines everywhere, several inputs, gzip and bgzip containers, FASTQ and FASTA mode, small blocks.  CPU only."""
import gzip
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HULK_INGEST_BLOCK", "131072")
import numpy as np
from hulk_amd import ingest
from hulk_amd._lib import HulkError
from oracle import linepump

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0


def soup(n_lines, clean):
    out = []
    i = 0
    while i < n_lines:
        r = rng.random()
        if clean or r < 0.80:                                   # a well-formed record
            L = int(rng.integers(1, 200))
            seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L))
            out += [b"@r%d" % i, seq, b"+", b"I" * L]
            i += 4
        elif r < 0.88:
            out.append(b""); i += 1
        elif r < 0.92:
            out.append(b">contig %d" % i); i += 1
        elif r < 0.96:
            out.append(bytes(rng.choice(np.frombuffer(b"ACGT@>+ \t", dtype=np.uint8), size=int(rng.integers(1, 30))))); i += 1
        else:
            out.append(b"@" * int(rng.integers(1, 4))); i += 1
    eol = [b"\n", b"\r\n"][int(rng.random() < 0.2)]
    data = eol.join(out)
    if rng.random() < 0.7:
        data += eol
    return data


def bgzf(data, eof_marker=True):
    import zlib
    out = bytearray()
    at, pieces = 0, []
    while at < len(data):
        n = int(rng.choice([1, 50, 700, 9000, 65280])); pieces.append(data[at:at + n]); at += n
    if eof_marker:
        pieces.append(b"")
    for piece in pieces:
        c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        total = 18 + len(body) + 8
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\x00BC\x02\x00" + (total - 1).to_bytes(2, "little")
        out += body + (zlib.crc32(piece) & 0xffffffff).to_bytes(4, "little") + len(piece).to_bytes(4, "little")
    return bytes(out)


with tempfile.TemporaryDirectory() as td:
    for case in range(n_cases):
        fasta = bool(rng.random() < 0.3)
        clean = bool(rng.random() < 0.4)
        paths = []
        for f in range(int(rng.integers(1, 4))):
            n_lines = int(rng.choice([0, 3, 17, 200, 4000]))
            data = soup(n_lines, clean)
            gz = bool(rng.random() < 0.3)
            p = os.path.join(td, "c%d_%d.fq%s" % (case, f, ".gz" if gz else ""))
            if gz and rng.random() < 0.4:                     # bgzip's container (members located by their BC field, inflated side by side),
                with open(p, "wb") as fh:                        # sometimes with an ordinary member in the middle (hand-over to the one-thread reader)
                    cut = int(rng.integers(0, len(data) + 1)) if rng.random() < 0.3 else len(data)
                    fh.write(bgzf(data[:cut], eof_marker=cut == len(data)))
                    if cut < len(data):
                        cut2 = int(rng.integers(cut, len(data) + 1))
                        fh.write(gzip.compress(data[cut:cut2], 1) + bgzf(data[cut2:]))
            else:
                with (gzip.open(p, "wb") if gz else open(p, "wb")) as fh:
                    fh.write(data)
            paths.append(p)
        want = werr = got = gerr = None
        try:
            want = [s if s is not None else b"" for s in linepump.sequences(paths, fasta)]
        except linepump.PumpError as e:
            werr = str(e)
        try:
            b, o, st = ingest.parse_files(paths, fasta=fasta, threads=int(rng.choice([1, 2, 5])))
            got = [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]
        except HulkError as e:
            gerr = e.message
        if want != got or werr != gerr:
            bad += 1
            print("MISMATCH case", case, "fasta" if fasta else "fastq", paths, "::", werr, "|", gerr,
                  "|", None if want is None else len(want), None if got is None else len(got), flush=True)
        for p in paths:
            os.unlink(p)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
