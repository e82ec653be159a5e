"""Native host ingest (hulk_parse_files: no GPU needed) against the literal restatement of the
reference's DataStreamer/FastqHandler (oracle/linepump.py, src/pipeline/sketch.go:40-161)."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from hulk_amd import ingest
from hulk_amd._lib import HulkError
from oracle import linepump
from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def native(paths, fasta=False, threads=0):
    b, o, st = ingest.parse_files(paths, fasta=fasta, threads=threads)
    seqs = [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]
    assert st["n_seqs"] == len(seqs) and st["total_len"] == sum(map(len, seqs))
    return seqs


def restated(paths, fasta=False):
    return [s if s is not None else b"" for s in linepump.sequences(paths, fasta)]


def write(tmp_path, name, data):
    p = str(tmp_path / name)
    if name.endswith(".gz"):
        with gzip.open(p, "wb") as fh:
            fh.write(data)
    else:
        with open(p, "wb") as fh:
            fh.write(data)
    return p


def test_reference_fixture():
    p = os.path.join(GOLDEN, "test-reads-small.fq.gz")
    got = native([p])
    assert got == restated([p]) and len(got) == 1000 and all(len(s) == 100 for s in got)


FASTQ_CASES = {
    "plain": b"@r1\nACGT\n+\nIIII\n@r2\nGGCC\n+\nIIII\n",
    "no_final_newline": b"@r1\nACGT\n+\nIIII\n@r2\nGGCC\n+\nIIII",
    "crlf": b"@r1\r\nACGT\r\n+\r\nIIII\r\n@r2\r\nGGCC\r\n+\r\nIIII\r\n",
    "empty_lines_skipped": b"\n\n@r1\n\nACGT\n\n\n+\nIIII\n\n@r2\nGGCC\n+\nIIII\n",
    "empty_fourth_line_completes": b"@r1\nACGT\n+\n\n@r2\nGGCC\n+\nIIII\n",
    "truncated_last_record": b"@r1\nACGT\n+\nIIII\n@r2\nGGCC\n+\n",
    "truncated_after_seq": b"@r1\nACGT\n+\nIIII\n@r2\nGGCC\n",
    "truncated_after_header": b"@r1\nACGT\n+\nIIII\n@r2\n",
    "bad_header_in_truncated_record": b"@r1\nACGT\n+\nIIII\nr2\nGGCC\n+\n",
    "cr_only_line_is_empty": b"@r1\nACGT\n+\nIIII\n\r\n@r2\nGGCC\n+\nIIII\n",
    "lowercase_and_n": b"@r1\nacgtNNnn\n+\nIIIIIIII\n",
    "empty": b"",
    "only_newlines": b"\n\n\n",
}


@pytest.mark.parametrize("name", sorted(FASTQ_CASES))
@pytest.mark.parametrize("gz", [False, True])
def test_fastq_cases(tmp_path, name, gz):
    p = write(tmp_path, "x.fq.gz" if gz else "x.fq", FASTQ_CASES[name])
    assert native([p]) == restated([p])


def test_fastq_bad_id_message(tmp_path):
    p = write(tmp_path, "bad.fq", b"@r1\nACGT\n+\nIIII\nr2 oops\nGGCC\n+\nIIII\n@r3\nAAAA\n+\nIIII\n")
    with pytest.raises(linepump.PumpError) as e0:
        restated([p])
    with pytest.raises(HulkError) as e1:
        native([p])
    assert e1.value.message == str(e0.value) == "read ID in fastq file does not begin with @: r2 oops"
    assert e1.value.code == -11


def test_token_too_long(tmp_path):
    ok = b"@r\n" + b"A" * 65535 + b"\n+\n" + b"I" * 65535 + b"\n"
    p = write(tmp_path, "ok.fq", ok)
    assert native([p]) == restated([p]) and len(native([p])[0]) == 65535
    bad = b"@r\n" + b"A" * 65536 + b"\n+\nI\n"
    p2 = write(tmp_path, "bad.fq", bad)
    with pytest.raises(linepump.PumpError):
        restated([p2])
    with pytest.raises(HulkError) as e:
        native([p2])
    assert e.value.message == "bufio.Scanner: token too long" and e.value.code == -12
    p3 = write(tmp_path, "nonl.fq", b"A" * 300000)          # no newline at all
    with pytest.raises(HulkError) as e:
        native([p3])
    assert e.value.code == -12


def test_multiple_files_are_one_line_stream(tmp_path):
    # a record may straddle inputs; an unterminated last line is a line of ITS input
    a = write(tmp_path, "a.fq", b"@r1\nACGT\n+\nIIII\n@r2\nGG")
    b = write(tmp_path, "b.fq.gz", b"+\nIIII\n@r3\nTTTT\n+\nIIII\n")
    got = native([a, b])
    assert got == restated([a, b]) == [b"ACGT", b"GG", b"TTTT"]   # "GG" ends a.fq as its own line
    c = write(tmp_path, "c.fq", b"@r4\nCCCC\n+\nIIII\n")
    assert native([a, b, c]) == restated([a, b, c])


def test_open_and_gzip_errors(tmp_path):
    with pytest.raises(HulkError) as e:
        native([str(tmp_path / "missing.fq")])
    assert e.value.code == -35 and "no such file" in e.value.message.lower()
    p = write(tmp_path, "notgz.fq", b"@r\nACGT\n+\nIIII\n")
    fake = str(tmp_path / "fake.fq.gz")
    os.rename(p, fake)
    with pytest.raises(HulkError) as e:
        native([fake])
    assert e.value.message == "gzip: invalid header"


FASTA_CASES = {
    "two_records": b">a desc\nACGT\nGGCC\n>b\nTTTT\n",
    "no_final_newline": b">a\nACGT\nGG",
    "stops_at_empty_line": b">a\nACGT\n\n>b\nTTTT\n",
    "lines_before_first_header_dropped": b"ACGT\n>a\nGGGG\n",
    "empty_record": b">a\n>b\nACGT\n",
    "crlf": b">a\r\nAC\r\nGT\r\n",
    "header_only": b">a\n",
}


@pytest.mark.parametrize("name", sorted(FASTA_CASES))
def test_fasta_cases(tmp_path, name):
    p = write(tmp_path, "x.fa", FASTA_CASES[name])
    assert native([p], fasta=True) == restated([p], fasta=True)


def test_fasta_without_header_is_an_error(tmp_path):
    p = write(tmp_path, "x.fa", b"ACGT\nGGCC\n")
    with pytest.raises(linepump.PumpError):
        restated([p], fasta=True)
    with pytest.raises(HulkError) as e:
        native([p], fasta=True)
    assert e.value.code == -36


def _random_fastq(rng, n, quirks):
    out = []
    for i in range(n):
        L = int(rng.integers(30, 300))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=L))
        rec = [b"@read%d" % i, seq, b"+", b"I" * L]
        if quirks:
            r = rng.random()
            if r < 0.02:
                rec.insert(int(rng.integers(0, 3)), b"")       # empty line before l1, l2 or l3: skipped
            elif r < 0.03:
                rec[3] = b""                                     # empty quality line: completes the record
            elif r < 0.04:
                rec = [x + b"\r" for x in rec]
        out.append(b"\n".join(rec) + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_many_blocks_and_pieces(tmp_path, threads):
    """~6 MB of FASTQ with line-level quirks, parsed with 128 KiB blocks so that records, empty lines
    and the 4-state machine cross hundreds of block and piece borders."""
    rng = np.random.default_rng(7)
    data = _random_fastq(rng, 18000, quirks=True)
    p = write(tmp_path, "big.fq", data)
    want = restated([p])
    code = ("import sys; sys.path.insert(0, %r); from hulk_amd import ingest; import hashlib;"
            "b, o, st = ingest.parse_files([%r], threads=%d);"
            "print(st['n_seqs'], hashlib.sha256(b.tobytes()).hexdigest(), hashlib.sha256(o.tobytes()).hexdigest())"
            % (ROOT, p, threads))
    env = dict(os.environ, HULK_INGEST_BLOCK="131072")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import hashlib
    n, hb, ho = out.stdout.split()
    offs = np.zeros(len(want) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in want], out=offs[1:])
    assert int(n) == len(want)
    assert hb == hashlib.sha256(b"".join(want)).hexdigest()
    assert ho == hashlib.sha256(offs.tobytes()).hexdigest()


def test_misframed_stream_reports_the_same_first_error(tmp_path):
    """An empty line between '+' and the qualities is taken as l4, so the quality line becomes the next
    header: both sides must stop at the same record with the same message, whatever the piece borders."""
    rng = np.random.default_rng(11)
    good = _random_fastq(rng, 3000, quirks=False)
    bad = b"@x\nACGT\n+\n\nIIII\n" + _random_fastq(rng, 3000, quirks=False)
    p = write(tmp_path, "mis.fq", good + bad)
    with pytest.raises(linepump.PumpError) as e0:
        restated([p])
    code = ("import sys; sys.path.insert(0, %r); from hulk_amd import ingest\n"
            "try:\n    ingest.parse_files([%r], threads=4)\nexcept Exception as e:\n    print(e.message)" % (ROOT, p))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, HULK_INGEST_BLOCK="131072"), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip() == str(e0.value) == "read ID in fastq file does not begin with @: IIII"


def test_stdin(tmp_path):
    data = FASTQ_CASES["empty_lines_skipped"]
    code = ("import sys; sys.path.insert(0, %r); from hulk_amd import ingest;"
            "b, o, st = ingest.parse_files([]); print(bytes(b).decode(), list(map(int, o)))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], input=data, capture_output=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.decode().strip() == "ACGTGGCC [0, 4, 8]"


def test_fuzz_slice():
    """Fixed-seed slice of tools/fuzz_ingest.py (random line soups, CR/LF, several inputs, gzip, both modes)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ingest.py"), "120", "5"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


def test_default_blocks_parallel_reads(tmp_path):
    """Default 32 MB blocks: a plain file is pread() in pieces side by side (ByteSource::read_pieces); two inputs,
    the first without a final newline and neither a multiple of the piece size — the reads must come out in
    order and complete."""
    rng = np.random.default_rng(11)
    paths, want = [], []
    for f, n in enumerate((190_000, 70_001)):
        lens = rng.integers(30, 251, n)
        tot = int(lens.sum())
        flat = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, tot)]
        offs = np.concatenate([[0], np.cumsum(lens)])
        chunks = []
        for i in range(n):
            s = flat[offs[i]:offs[i + 1]].tobytes()
            want.append(s)
            chunks.append(b"@%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))
        data = b"".join(chunks)
        if f == 0:
            data = data[:-1]
        assert len(data) > (32 << 20) or f == 1
        paths.append(write(tmp_path, "big%d.fq" % f, data))
    got = native(paths)
    assert len(got) == len(want) and got == want


# ---- gzip input through hulk::inflate (fast_inflate.h) ----

def _fastq_blob(rng, n, lo=30, hi=251):
    lens = rng.integers(lo, hi, n)
    seqs = [bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(L))]) for L in lens]
    return seqs, b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"F" * len(s)) for i, s in enumerate(seqs))


def test_gzip_large_levels_and_members(tmp_path):
    """Several MB of FASTQ (many 1 MB inflate chunks, matches that reach across them) at compression levels 1 / 6 / 9,
    as one member and as three concatenated members (compress/gzip reads multistream), with FNAME/FEXTRA/FCOMMENT
    header fields; the same through zlib's inflate (HULK_GZ_ZLIB=1) in a child process."""
    import zlib
    rng = np.random.default_rng(5)
    seqs, blob = _fastq_blob(rng, 40_000)
    assert len(blob) > 10 << 20

    def member(data, level, flags=0):
        hdr = bytearray(b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\0\xff")
        if flags & 4: hdr += b"\x05\x00hello"
        if flags & 8: hdr += b"name.fq\0"
        if flags & 16: hdr += b"a comment\0"
        if flags & 2: hdr += b"\x12\x34"
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(data) + c.flush()
        return bytes(hdr) + body + (zlib.crc32(data) & 0xffffffff).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")

    paths = []
    for level in (1, 6, 9):
        p = str(tmp_path / ("one%d.fq.gz" % level))
        open(p, "wb").write(member(blob, level, flags=4 | 8 | 16 | 2))
        assert native([p]) == seqs
        paths.append(p)
    cut1, cut2 = blob.index(b"\n@r9000\n") + 1, blob.index(b"\n@r30000\n") + 1
    p3 = str(tmp_path / "three.fq.gz")
    open(p3, "wb").write(member(blob[:cut1], 6) + member(blob[cut1:cut2], 1, flags=8) + member(blob[cut2:], 9) + b"\0\0trailing garbage")
    assert native([p3]) == seqs
    # stored blocks (level 0) and an empty member in the middle
    p0 = str(tmp_path / "stored.fq.gz")
    open(p0, "wb").write(member(blob[:cut1], 0) + member(b"", 6) + member(blob[cut1:], 0))
    assert native([p0]) == seqs
    code = ("import sys; sys.path.insert(0, %r)\nfrom hulk_amd import ingest\n"
            "b, o, st = ingest.parse_files(%r)\nimport hashlib; print(st['n_seqs'], hashlib.md5(b.tobytes()).hexdigest())" % (ROOT, [p3, p0]))
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **extra)).stdout.strip() for extra in ({}, {"HULK_GZ_ZLIB": "1"})]
    assert outs[0] == outs[1] and outs[0].startswith(str(2 * len(seqs)) + " ")


def test_gzip_errors(tmp_path):
    import zlib
    rng = np.random.default_rng(6)
    seqs, blob = _fastq_blob(rng, 3000)
    good = gzip.compress(blob, 6)
    bad_crc = bytearray(good); bad_crc[-5] ^= 0x40
    p = write(tmp_path, "crc.fq", b""); p = str(tmp_path / "crc.fq.gz"); open(p, "wb").write(bytes(bad_crc))
    with pytest.raises(HulkError) as e:
        native([p])
    assert e.value.message == "gzip: invalid checksum"
    bad_size = bytearray(good); bad_size[-1] ^= 0x01
    p = str(tmp_path / "size.fq.gz"); open(p, "wb").write(bytes(bad_size))
    with pytest.raises(HulkError) as e:
        native([p])
    assert e.value.message == "gzip: invalid checksum"
    p = str(tmp_path / "cut.fq.gz"); open(p, "wb").write(good[:len(good) // 2])
    with pytest.raises(HulkError) as e:
        native([p])
    assert e.value.message == "unexpected EOF"
    p = str(tmp_path / "cuttrailer.fq.gz"); open(p, "wb").write(good[:-3])
    with pytest.raises(HulkError) as e:
        native([p])
    assert e.value.message == "unexpected EOF"
    # a flipped bit in the deflate data: a flate error or a checksum error, never silent acceptance
    for at in (40, len(good) // 3, len(good) // 2):
        mangled = bytearray(good); mangled[at] ^= 0x10
        p = str(tmp_path / ("flip%d.fq.gz" % at)); open(p, "wb").write(bytes(mangled))
        with pytest.raises(HulkError) as e:
            native([p])
        assert e.value.message.startswith("gzip: ") or e.value.message in ("unexpected EOF",) or "FASTQ" in e.value.message or "@" in e.value.message
    p = str(tmp_path / "method.fq.gz"); open(p, "wb").write(b"\x1f\x8b\x07" + good[3:])
    with pytest.raises(HulkError) as e:
        native([p])
    assert e.value.message == "gzip: invalid header"


def test_inflate_against_zlib(tmp_path):
    """tests/cpp/inflate_fuzz.cpp: the decoder against zlib on random data, all levels/strategies, random feed sizes."""
    exe = str(tmp_path / "inflate_fuzz")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "inflate_fuzz.cpp"), "-lz"],
                   check=True, timeout=300)
    r = subprocess.run([exe, "250", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]


def _bgzf(data, rng, level=6, eof_marker=True, extra_subfield=False):
    """bgzip's container: members of <= 64 KiB of text, each header with the 'BC' extra subfield = member size - 1
    (optionally behind another subfield), and the empty end-of-file member."""
    import zlib
    out = bytearray()
    at = 0
    pieces = []
    while at < len(data):
        n = int(rng.integers(1, 65281)) if rng.random() < 0.3 else 65280
        pieces.append(data[at:at + n]); at += n
    if eof_marker:
        pieces.append(b"")
    for piece in pieces:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        pre = b"XY\x03\x00abc" if extra_subfield else b""
        xlen = len(pre) + 6
        total = 12 + xlen + len(body) + 8
        assert total <= 65536
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + xlen.to_bytes(2, "little") + pre + b"BC\x02\x00" + (total - 1).to_bytes(2, "little")
        out += body + (zlib.crc32(piece) & 0xffffffff).to_bytes(4, "little") + len(piece).to_bytes(4, "little")
    return bytes(out)


def test_bgzf_members_are_inflated_side_by_side(tmp_path):
    """bgzip files go through the parallel member reader (GzBgzf): same reads as the text; an ordinary gzip member, trailing
    bytes, a lying BSIZE, a bad checksum or a cut file in the middle hand the stream over to the one-thread reader, whose
    results and messages they must therefore keep; the one-thread reader alone (HULK_GZ_THREADS=1) agrees."""
    rng = np.random.default_rng(11)
    seqs, blob = _fastq_blob(rng, 60_000)                       # ~ 17 MB of text: several 8 MB pieces of compressed input? (one or two) and ~270 members
    assert len(blob) > 16 << 20
    p = str(tmp_path / "a.fq.gz"); open(p, "wb").write(_bgzf(blob, rng))
    assert native([p]) == seqs
    p = str(tmp_path / "sub.fq.gz"); open(p, "wb").write(_bgzf(blob, rng, level=1, extra_subfield=True, eof_marker=False))
    assert native([p]) == seqs
    # BGZF, then an ordinary member, then BGZF again, then bytes that are no gzip header
    cut1, cut2 = blob.index(b"\n@r20000\n") + 1, blob.index(b"\n@r41000\n") + 1
    mixed = _bgzf(blob[:cut1], rng, eof_marker=False) + gzip.compress(blob[cut1:cut2], 6) + _bgzf(blob[cut2:], rng) + b"\0\0not gzip"
    p = str(tmp_path / "mixed.fq.gz"); open(p, "wb").write(mixed)
    assert native([p]) == seqs
    good = _bgzf(blob, rng)
    # BSIZE of the third member off by a few bytes: the members cannot be located from there on, the stream itself is fine
    hdr = [i for i in range(len(good) - 18) if good[i:i + 4] == b"\x1f\x8b\x08\x04" and good[i + 12:i + 16] == b"BC\x02\x00"]
    lying = bytearray(good); at = hdr[2] + 16
    lying[at:at + 2] = (int.from_bytes(good[at:at + 2], "little") + 7).to_bytes(2, "little")
    p = str(tmp_path / "lying.fq.gz"); open(p, "wb").write(bytes(lying))
    assert native([p]) == seqs
    # checksum of a member in the middle / a cut file: the reference's messages
    bad = bytearray(good); bad[hdr[5] - 6] ^= 0x20                # (CRC-32 field of member 4)
    p = str(tmp_path / "crc.fq.gz"); open(p, "wb").write(bytes(bad))
    with pytest.raises(HulkError) as e:
        native([p])
    assert e.value.message == "gzip: invalid checksum"
    p = str(tmp_path / "cut.fq.gz"); open(p, "wb").write(good[:hdr[len(hdr) // 2] + 300])
    with pytest.raises(HulkError) as e:
        native([p])
    assert e.value.message == "unexpected EOF"
    # a file that is only the end-of-file member: no reads, no error
    p = str(tmp_path / "empty.fq.gz"); open(p, "wb").write(_bgzf(b"", rng))
    assert native([p]) == []
    code = ("import sys; sys.path.insert(0, %r)\nfrom hulk_amd import ingest\n"
            "b, o, st = ingest.parse_files(%r)\nimport hashlib; print(st['n_seqs'], hashlib.md5(b.tobytes()).hexdigest())"
            % (ROOT, [str(tmp_path / "a.fq.gz"), str(tmp_path / "mixed.fq.gz"), str(tmp_path / "lying.fq.gz")]))
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **extra)).stdout.strip() for extra in ({}, {"HULK_GZ_THREADS": "1"}, {"HULK_GZ_ZLIB": "1"})]
    assert outs[0] == outs[1] == outs[2] and outs[0].startswith(str(3 * len(seqs)) + " ")


def _gz_children(paths, **env):
    """tools/fuzz_gzpar.py --child in a fresh process (the gzip readers read their switches once): {path: [n_seqs, md5(bases),
    md5(offsets)] or ["error", message]} and the parallel reader's trace lines"""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gzpar.py"), "--child"] + paths, capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, HULK_INGEST_TRACE="1", **env))
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("[")]
    assert len(rows) == len(paths), r.stderr[-2000:]
    return {row[0]: row[1:] for row in rows}, [l for l in r.stderr.splitlines() if "parallel gzip reader," in l]


def test_one_gzip_member_is_inflated_by_several_threads(tmp_path):
    """An ordinary `.gz` (one member) goes through GzPar: chunks decoded speculatively from block starts found in the stream,
    counted only when the chunk in front ends exactly there, unknown window bytes resolved afterwards, CRC-32 combined — and the
    final block, the trailer, further members, trailing bytes and every error are the one-thread reader's.  Same reads as the
    text at levels 1 / 6 / 9 (small chunks: dozens of them, several batches); a second member and trailing bytes; a wrong
    CRC-32, a wrong ISIZE, a cut file and a flipped bit give the one-thread reader's messages (HULK_GZ_PAR=0)."""
    import hashlib
    import zlib
    rng = np.random.default_rng(21)
    seqs, blob = _fastq_blob(rng, 40_000)
    want = [len(seqs), hashlib.md5(b"".join(seqs)).hexdigest()]
    paths = []
    for level in (1, 6, 9):
        p = str(tmp_path / ("one%d.fq.gz" % level)); open(p, "wb").write(gzip.compress(blob, level)); paths.append(p)
    cut = blob.index(b"\n@r25000\n") + 1
    p2 = str(tmp_path / "two.fq.gz"); open(p2, "wb").write(gzip.compress(blob[:cut], 6) + gzip.compress(blob[cut:], 1) + b"\0\0 not gzip")
    # stored blocks only: no block start is ever found, the chain is one chunk per batch — and the reader gives up
    c0 = zlib.compressobj(0, zlib.DEFLATED, 31); p0 = str(tmp_path / "stored.fq.gz"); open(p0, "wb").write(c0.compress(blob) + c0.flush())
    good = gzip.compress(blob, 6)
    bad = {}
    for name, f in (("crc", lambda d: d[:-5] + bytes([d[-5] ^ 0x40]) + d[-4:]), ("isize", lambda d: d[:-1] + bytes([d[-1] ^ 1])),
                    ("cut", lambda d: d[:len(d) // 2]), ("cuttrailer", lambda d: d[:-3]),
                    ("flip", lambda d: d[:len(d) // 2] + bytes([d[len(d) // 2] ^ 0x10]) + d[len(d) // 2 + 1:])):
        p = str(tmp_path / (name + ".fq.gz")); open(p, "wb").write(f(good)); bad[name] = p
    every = paths + [p2, p0] + list(bad.values())
    par, trace = _gz_children(every, HULK_GZ_PAR_CHUNK="65536", HULK_GZ_THREADS="3")
    one, none = _gz_children(every, HULK_GZ_PAR="0")
    assert none == [] and len(trace) == len(every)                          # the parallel reader ran on every file, and only when asked
    counted = [int(l.split(" chunks counted")[0].split()[-1]) for l in trace]
    assert min(counted[:4]) >= 20, trace                                       # ... and did the work (dozens of chunks counted per file)
    for p in paths + [p2, p0]:
        assert par[p][:2] == want and one[p] == par[p], (p, par[p], one[p])
    assert par[bad["crc"]] == par[bad["isize"]] == ["error", "gzip: invalid checksum"]
    assert par[bad["cut"]] == par[bad["cuttrailer"]] == ["error", "unexpected EOF"]
    for p in bad.values():
        assert par[p] == one[p], (p, par[p], one[p])
    # default chunk size (1 MB): a file of fewer than four chunks is the one-thread reader's, a larger one is not
    big = str(tmp_path / "big.fq.gz"); open(big, "wb").write(gzip.compress(blob * 2, 1))
    assert os.path.getsize(paths[1]) < 4 << 20 < os.path.getsize(big)
    dflt, trace = _gz_children([paths[1], big])
    assert len(trace) == 1 and dflt[paths[1]][:2] == want and dflt[big][0] == 2 * len(seqs)


def test_gzpar_fuzz_slice():
    """tools/fuzz_gzpar.py: both gzip readers on random members (levels, strategies, flush points, several members, flipped
    bits, truncation, binary stretches) — the same reads or the same message."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gzpar.py"), "12", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_large_piece_copy_covers_its_last_bytes(tmp_path):
    """A piece of >= 8 MB is copied out of an inflate batch by four threads.  With pieces of floor(n / 4) bytes (as first
    written) the last n mod 4 bytes stayed uncopied whenever floor(n / 4) was a multiple of 64: here n = 4 * 64 * 40000 + 2,
    and the two bytes are the last base of a FASTA record and its newline (found by the parallel gzip reader's first test)."""
    rng = np.random.default_rng(31)
    n = 4 * 64 * 40000 + 2
    head = b">contig\n"
    raw = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n - len(head))].tobytes())
    raw[70::71] = b"\n" * len(raw[70::71])                                     # lines of 70 bases
    raw[-1:] = b"\n"
    assert raw[-2:-1] != b"\n"
    text = head + bytes(raw)
    body = bytes(raw).replace(b"\n", b"")
    assert len(text) == n and (n // 4) % 64 == 0 and n % 4 == 2
    p = str(tmp_path / "exact.fa.gz"); open(p, "wb").write(_bgzf(text, rng, level=1))
    assert os.path.getsize(p) < 8 << 20                                       # one batch of members, delivered by one read
    assert native([p], fasta=True) == [body]


def test_knobs_are_fields_of_the_run_not_of_the_process(tmp_path):
    """hulk_ingest_opts (ABI 3): block size, gzip threads / chunk, the one-thread and zlib readers are fields of ONE call —
    several configurations in this one process, no environment variable, the reads of the restated line pump every time;
    unknown flags are refused."""
    from hulk_amd import _lib, ingest
    rng = np.random.default_rng(404)
    data = _random_fastq(rng, 9000, quirks=True)
    plain = write(tmp_path, "k.fq", data)
    gz = write(tmp_path, "k.fq.gz", data)                       # (write() compresses what ends in .gz)
    want = restated([plain])
    body = b"".join(want)
    for path, opts in ((plain, {"block_bytes": 131072}), (plain, {"block_bytes": 131072, "file_readers": 1, "parser_threads": 3}),
                       (gz, {"gz_threads": 3, "gz_chunk_bytes": 65536, "block_bytes": 262144}),
                       (gz, {"flags": _lib.HULK_INGEST_GZ_ONE_THREAD}), (gz, {"flags": _lib.HULK_INGEST_GZ_ZLIB, "block_bytes": 131072}),
                       (gz, {"parser_threads": 40})):
        b, o, st = ingest.parse_files([path], opts=opts)
        assert st["n_seqs"] == len(want) and b.tobytes() == body, opts
    with pytest.raises(_lib.HulkError):
        ingest.parse_files([plain], opts={"flags": 1 << 9})


def test_ingest_opts_are_validated_like_hulk_params(tmp_path):
    """Out-of-range fields, unknown flags and non-zero reserved words of hulk_ingest_opts are refused (HULK_ERR_ARG, with the
    field's name), not clamped — as hulk_create treats hulk_params."""
    import ctypes
    from hulk_amd import _lib, ingest
    plain = write(tmp_path, "v.fq", b"@r\nACGT\n+\nIIII\n")
    for opts, word in (({"gz_threads": 65}, "gz_threads"), ({"file_readers": 17}, "file_readers"), ({"parser_threads": 257}, "parser_threads"),
                       ({"block_bytes": 4096}, "block_bytes"), ({"gz_chunk_bytes": 100}, "gz_chunk_bytes"), ({"flags": 1 << 12}, "flags")):
        with pytest.raises(_lib.HulkError) as e:
            ingest.parse_files([plain], opts=opts)
        assert e.value.code == -30 and word in str(e.value), (opts, str(e.value))
    o = _lib.IngestOpts()
    o.reserved[1] = 7
    L = _lib.load()
    err = ctypes.create_string_buffer(256)
    arr, n = ingest._path_array([plain])
    rc = L.hulk_parse_files_opts(arr, n, 0, ctypes.byref(o), ctypes.cast(None, _lib.BATCH_FN), None, None, err, 256)
    assert rc == -30 and b"reserved" in err.value
    # the host-parser flag is a known flag (it only matters to hulk_sketch_files)
    b, off, st = ingest.parse_files([plain], opts={"flags": _lib.HULK_INGEST_HOST_PARSER})
    assert st["n_seqs"] == 1 and b.tobytes() == b"ACGT"


@pytest.mark.parametrize("threads,block", [(1, 0), (2, 131072), (5, 131072), (16, 262144), (16, 0)])
def test_fasta_parallel_pieces_equal_the_restatement(tmp_path, threads, block):
    """--fasta (sketch.go:102-135) with a block cut into pieces parsed side by side (round 6): records that span pieces and blocks,
    sequence lines in front of the first header (dropped), CR/LF, headers back to back (an empty record), a header as the last
    line, and an EMPTY line that ends the parsing — in the first piece, in a later one, right behind a header — against the
    literal restatement, for several thread counts and block sizes (128 KiB blocks: a 700 kb record spans six of them)."""
    rng = np.random.default_rng(threads * 1000 + block // 1024)
    acgt = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)

    def record(name, L, width, eol=b"\n"):
        seq = bytes(acgt[rng.integers(0, len(acgt), size=L)])
        return b">" + name + eol + b"".join(seq[i:i + width] + eol for i in range(0, L, width))

    body = (b"ACGTACGT\nTTTT\n" +                                  # no record owns these
            record(b"c1 first", 700_000, 60) + record(b"c2", 1, 60) + b">empty_record\n" + b">c3\n" + record(b"c4 crlf", 250_000, 70, b"\r\n") +
            record(b"c5", 333_333, 61) + b">last_header_without_sequence\n")
    opts = {"block_bytes": block} if block else None

    def nat(path):
        b, o, st = ingest.parse_files([path], fasta=True, threads=threads, opts=opts)
        return [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)], st

    p = write(tmp_path, "full.fa", body)
    got, st = nat(p)
    want = restated([p], fasta=True)
    assert got == want and len(got) == 7 and [len(x) for x in got] == [700_000, 1, 0, 0, 250_000, 333_333, 0]
    assert st["n_lines"] == body.count(b"\n")
    # an empty line ends the parsing wherever it stands; nothing behind it counts (not even a line of 64 KiB)
    for cut in (5, 12_345, 400_000, 711_700, len(body) // 2, len(body) - 40):
        at = body.index(b"\n", cut) + 1
        q = write(tmp_path, f"stop_{cut}.fa", body[:at] + b"\n" + b"A" * 70_000 + b"\n" + body[at:])
        try:
            want = restated([q], fasta=True)
        except linepump.PumpError:                                  # the empty line stands in front of the first header: the reference dies on l1[0] = 64
            with pytest.raises(HulkError, match="no header"):
                nat(q)
            continue
        got, st = nat(q)
        assert got == want, cut
        assert st["n_lines"] == body[:at].count(b"\n") + 1
    # a line of >= 64 KiB in front of any empty line is the scanner's error (sketch.go:53: bufio.Scanner: token too long)
    q = write(tmp_path, "long.fa", body[:300_000] + b"C" * 65_536 + b"\n" + body[300_000:])
    with pytest.raises(HulkError, match="token too long"):
        nat(q)


def test_recycled_buffers_carry_nothing_over(tmp_path):
    """The ingest path's large buffers (block, piece and batch regions) go back to a pool of the PROCESS and are handed out again
    with their old contents (hulk_ingest.hip RegionPool).  A long FASTA file, then a shorter one with other bases, then FASTQ,
    with and without hulk_release_caches() in between: every parse equals the restatement."""
    from hulk_amd import _lib
    rng = np.random.default_rng(77)

    def fa(n_rec, L, alphabet):
        a = np.frombuffer(alphabet, dtype=np.uint8)
        out = []
        for i in range(n_rec):
            seq = bytes(a[rng.integers(0, len(a), size=L)])
            out.append(b">r%d\n" % i + b"".join(seq[j:j + 60] + b"\n" for j in range(0, L, 60)))
        return b"".join(out)

    long_ = write(tmp_path, "long.fa", fa(3, 500_000, b"ACGT"))
    short = write(tmp_path, "short.fa", fa(5, 150_001, b"TG"))
    fq = write(tmp_path, "r.fq", b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGTTGCA" * 20, b"I" * 160) for i in range(5000)))
    want = {p_: restated([p_], fasta=f_) for p_, f_ in ((long_, True), (short, True), (fq, False))}
    for release in (False, True, False):
        for path, fasta in ((long_, True), (short, True), (fq, False), (short, True), (long_, True)):
            b, o, st = ingest.parse_files([path], fasta=fasta, threads=8)
            got = [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]
            assert got == want[path], (path, release)
            if release:
                assert _lib.load().hulk_release_caches() == 0
