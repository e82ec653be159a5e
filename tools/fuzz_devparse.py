#!/usr/bin/env python3
"""Randomised differential test of the DEVICE FASTQ parser (hulk_fastq.hip behind hulk_sketch_files) against the literal
restatement of the reference's stream pump (oracle/linepump.py): random line soups — well-formed records, empty lines
everywhere, CR/LF mixes, stray '>' and '@' lines, missing final newlines, several inputs, gzip — in small blocks, so that
records straddle block borders in every phase.  What must agree with the oracle: the error text, or the number of reads,
their total length, the number of lines, and the k-mer spectrum of everything binned (the spectrum of hulk_add_reads over
the oracle's reads; interval 0: one spectrum for the run).  The host parser (HULK_INGEST_HOST_PARSER) runs beside it.
FUZZ_FASTA=1: the same for --fasta (k_fa_* behind hulk_sketch_files(fasta = 1), sketch.go:102-135): sequence lines of any length, '>'
lines anywhere (back to back, first, last), sequence lines in front of the first header, an empty line somewhere (it ends the
parsing), CR/LF, several inputs that continue one another's record.
usage: fuzz_devparse.py [n_cases] [seed]     (run on the GPU box)"""
import gzip
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before libhulkhip, see hulk_amd/_lib.py)
import hulk_amd
from hulk_amd import _lib
from hulk_amd._lib import HulkError
from oracle import linepump

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
K, W = 7, 3
MINLEN = W + K - 1
bad = 0
took_over = 0
FASTA = bool(os.environ.get("FUZZ_FASTA"))


def fasta_soup(n_lines, kind):
    """kind 0: clean records; 1: + junk in front, headers back to back; 2: + an empty line somewhere, short records"""
    out = []
    if kind >= 1 and rng.random() < 0.5:
        out += [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(1, 90)))) for _ in range(int(rng.integers(1, 4)))]
    while len(out) < n_lines:
        out.append(b">contig %d some text" % len(out) if rng.random() < 0.9 else b">")
        if kind >= 1 and rng.random() < 0.02:
            continue                                                        # (headers back to back: an empty record)
        Lmax = int(rng.choice([MINLEN, 40, 300, 5000, 200000]))
        short = kind == 2 and rng.random() < 0.03
        L = int(rng.integers(1, MINLEN)) if short else int(rng.integers(MINLEN, Lmax + 1))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), size=L))
        width = int(rng.choice([60, 70, 80, 1, 7, 1000, 65000]))
        out += [seq[i:i + width] for i in range(0, L, width)]
    if kind == 2 and rng.random() < 0.5 and out:
        out.insert(int(rng.integers(0, len(out) + 1)), b"")
    eol = [b"\n", b"\r\n"][int(rng.random() < 0.2)]
    data = eol.join(out)
    if rng.random() < 0.7:
        data += eol
    return data


def soup(n_lines, kind):
    """kind 0: clean records; 1: records + empty lines (legal for the slot machine); 2: anything"""
    out, i = [], 0
    while i < n_lines:
        r = rng.random()
        if kind == 0 or r < 0.80:
            L = int(rng.integers(MINLEN, 220)) if rng.random() < 0.97 or kind < 2 else int(rng.integers(1, MINLEN))
            seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L))
            out += [b"@r%d" % i, seq, b"+", b"I" * L if rng.random() < 0.9 else b""]     # (an empty 4th line completes a record too)
            i += 4
        elif r < 0.92 or kind == 1:
            out += [b""] * int(rng.choice([1, 1, 2, 5])); i += 1
        elif r < 0.95:
            out.append(b">contig %d" % i); i += 1
        elif r < 0.98:
            out.append(bytes(rng.choice(np.frombuffer(b"ACGT@>+ \t", dtype=np.uint8), size=int(rng.integers(1, 30))))); i += 1
        else:
            out.append(b"@" * int(rng.integers(1, 4))); i += 1
    eol = [b"\n", b"\r\n"][int(rng.random() < 0.2)]
    data = eol.join(out)
    if rng.random() < 0.7:
        data += eol
    return data


t_start = time.time()
budget = float(os.environ.get("FUZZ_SECONDS", 0))                    # stop after this many seconds (the summary counts the cases done)
with tempfile.TemporaryDirectory() as td:
    for case in range(n_cases):
        if budget and time.time() - t_start > budget:
            n_cases = case
            break
        kind = int(rng.choice([0, 0, 1, 1, 2]))
        paths = []
        for f in range(int(rng.integers(1, 4))):
            n_lines = int(rng.choice([0, 3, 17, 200, 4000, 30000]))
            data = fasta_soup(n_lines, kind) if FASTA else soup(n_lines, kind)
            if rng.random() < 0.03:                                  # a line of 64 KiB: "bufio.Scanner: token too long"
                at = int(rng.integers(0, len(data) + 1))
                data = data[:at] + b"A" * 65536 + data[at:]
            gz = bool(rng.random() < 0.2)
            p = os.path.join(td, "c%d_%d.fq%s" % (case, f, ".gz" if gz else ""))
            with (gzip.open(p, "wb", compresslevel=1) if gz else open(p, "wb")) as fh:
                fh.write(data)
            paths.append(p)
        block = int(rng.choice([131072, 131072, 262144, 1 << 20]))
        # the reference dies at the FIRST problem of the stream; the library checks a block's read lengths when the block is
        # handed over, i.e. it may meet a later line-level error of the same block first (the host parser always did): every
        # problem of the stream up to and including the first line-level one is an acceptable message
        want, acceptable = [], set()
        try:
            for s_ in linepump.fastq_handler(linepump.data_streamer(paths), FASTA):
                s_ = s_ if s_ is not None else b""
                if len(s_) == 0:
                    acceptable.add("sequence length must be > 0")
                elif len(s_) < MINLEN:
                    acceptable.add("sequence length must be >= w + k - 1")
                want.append(s_)
        except linepump.PumpError as e:
            acceptable.add(str(e))
        werr = sorted(acceptable) if acceptable else None
        if acceptable:
            want = None
        res = {}
        for label, flags in (("device", 0), ("host", _lib.HULK_INGEST_HOST_PARSER)):
            g = hulk_amd.GpuSketcher(K, W, 8, interval=0)
            try:
                st = g.sketch_files(paths, fasta=FASTA, opts={"flags": flags, "block_bytes": block, "parser_threads": int(rng.choice([1, 3]))})
                g.synchronize()
                res[label] = (None, (st["n_seqs"], st["total_len"], st["n_lines"]), g.histogram().copy())
            except HulkError as e:
                res[label] = (e.message, None, None)
            g.close()
        if want is not None:
            o = hulk_amd.GpuSketcher(K, W, 8, interval=0)
            if want:
                bases = np.frombuffer(b"".join(want), dtype=np.uint8)
                offsets = np.zeros(len(want) + 1, dtype=np.uint64)
                np.cumsum([len(s) for s in want], out=offsets[1:])
                o.add_reads(bases, offsets)
            o.synchronize()
            hist = o.histogram().copy()
            o.close()
            n_lines = 0
            for ln in linepump.data_streamer(paths):
                n_lines += 1
                if FASTA and ln is None:                                   # (the empty line that ends the parsing is the last one read)
                    break
            exp = (None, (len(want), sum(len(s) for s in want), n_lines), hist)
        else:
            exp = (werr, None, None)
        for label in ("device", "host"):
            got = res[label]
            if acceptable:
                same = got[0] in acceptable
            else:
                same = got[0] is None and got[1] == exp[1] and np.array_equal(got[2], exp[2])
            if not same:
                bad += 1
                print("MISMATCH case", case, label, "kind", kind, "block", block, paths, "::", exp[0], "|", got[0], "|", exp[1], got[1], flush=True)
        for p in paths:
            os.unlink(p)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t_start:.1f} s (seed {seed})")
sys.exit(1 if bad else 0)
