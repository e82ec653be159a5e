"""hulk_sketch_files on the GPU: the native ingest feeding the HIP path must give the sketch that
hulk_add_reads gives for the same reads, and the oracle's."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, pack_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu():
    import hulk_amd
    return hulk_amd


def test_fixture_file_matches_add_reads_and_oracle(fq_reads):
    from oracle import pyorc
    p = os.path.join(GOLDEN, "test-reads-small.fq.gz")
    a = gpu().GpuSketcher(21, 9, 128)
    st = a.sketch_files([p])
    a.finish()
    assert st["n_seqs"] == 1000 and st["total_len"] == 100000
    b = gpu().GpuSketcher(21, 9, 128)
    b.add_reads(*pack_reads(fq_reads))
    b.finish()
    ma, wa = a.sketch(); mb, wb = b.sketch()
    assert np.array_equal(ma, mb) and np.array_equal(wa, wb)
    assert a.counters() == b.counters()
    o = pyorc.Sketcher(21, 9, 128)
    for r in fq_reads:
        o.add_read(r)
    o.finish()
    mo, wo = o.sketch()
    assert np.array_equal(ma, mo) and np.allclose(wa, wo, rtol=1e-12, atol=0)
    a.close(); b.close(); o.close()


def test_intervals_across_many_blocks(tmp_path):
    """60k reads with an interval of 7000, parsed in 128 KiB blocks: batches, intervals and the two
    staging buffers interleave; result = one hulk_add_reads call over the same reads."""
    from hulk_amd import synth
    n, L = 60000, 150
    bases, offsets = synth.reads_numpy(0, n, L)
    raw = bases[:n * L].tobytes()
    p = str(tmp_path / "r.fq")
    with open(p, "wb") as fh:
        fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, raw[i * L:(i + 1) * L], b"I" * L) for i in range(n)))
    code = ("import sys; sys.path.insert(0, %r); import hulk_amd, hashlib\n"
            "g = hulk_amd.GpuSketcher(15, 9, 64, interval=7000)\n"
            "st = g.sketch_files([%r], threads=4); g.finish(); m, w = g.sketch()\n"
            "print(st['n_seqs'], hashlib.sha256(m.tobytes() + w.tobytes()).hexdigest())" % (ROOT, p))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, HULK_INGEST_BLOCK="131072"), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import hashlib
    g = gpu().GpuSketcher(15, 9, 64, interval=7000)
    g.add_reads(bases, offsets)
    g.finish()
    m, w = g.sketch()
    g.close()
    cnt, h = out.stdout.split()
    assert int(cnt) == n and h == hashlib.sha256(m.tobytes() + w.tobytes()).hexdigest()


def test_fasta_and_errors(tmp_path):
    from hulk_amd import synth
    from hulk_amd._lib import HulkError
    bases, _ = synth.reads_numpy(5, 3, 5000)
    seqs = [bases[i * 5000:(i + 1) * 5000].tobytes() for i in range(3)]
    p = str(tmp_path / "g.fa.gz")
    with gzip.open(p, "wb") as fh:
        for i, s in enumerate(seqs):
            fh.write(b">c%d\n" % i + b"\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + b"\n")
    a = gpu().GpuSketcher(11, 9, 32)
    st = a.sketch_files([p], fasta=True)
    a.finish()
    assert st["n_seqs"] == 3 and st["total_len"] == 15000
    b = gpu().GpuSketcher(11, 9, 32)
    b.add_reads(*pack_reads(seqs))
    b.finish()
    assert np.array_equal(a.sketch()[0], b.sketch()[0]) and np.array_equal(a.sketch()[1], b.sketch()[1])
    a.close(); b.close()
    # a read shorter than w + k - 1 is the reference's fatal error (minimizer.go:75)
    q = str(tmp_path / "short.fq")
    open(q, "wb").write(b"@r\nACGTACGT\n+\nIIIIIIII\n")
    c = gpu().GpuSketcher(21, 9, 32)
    with pytest.raises(HulkError) as e:
        c.sketch_files([q])
    assert e.value.message == "sequence length must be >= w + k - 1"
    c.close()
    d = gpu().GpuSketcher(21, 9, 32)
    bad = str(tmp_path / "bad.fq")
    open(bad, "wb").write(b"r1\nACGT\n+\nIIII\n")
    with pytest.raises(HulkError) as e:
        d.sketch_files([bad])
    assert e.value.message == "read ID in fastq file does not begin with @: r1"
    d.close()


def test_device_parser_host_parser_and_released_caches_agree(tmp_path):
    """hulk_sketch_files with the FASTQ line machine on the device (the default), on the host's parser threads
    (HULK_INGEST_HOST_PARSER) and again on the device after hulk_release_caches() dropped the process's buffers: the same
    reads, lines and sketch.  Records straddle 128 KiB blocks; CR/LF line ends, empty lines between records, an unterminated last
    line; a second input continues the slot machine of the first."""
    from hulk_amd import _lib, synth
    rng = np.random.default_rng(12)
    bases, _ = synth.reads_numpy(5, 30000, 150)
    raw = bases[:30000 * 150].tobytes()
    recs = []
    for i in range(30000):
        eol = b"\r\n" if i % 7 == 0 else b"\n"
        recs.append(b"@read%d some text" % i + eol + raw[i * 150:(i + 1) * 150] + eol + b"+" + eol + b"I" * 150 + eol)
        if i % 1000 == 3:
            recs.append(eol * int(rng.integers(1, 4)))                     # empty lines are skipped in front of a header
    half = len(recs) // 2
    p1, p2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    open(p1, "wb").write(b"".join(recs[:half]).rstrip(b"\r\n"))          # the first input's last line is not terminated
    open(p2, "wb").write(b"".join(recs[half:]))
    from oracle import linepump
    want = linepump.sequences([p1, p2])
    res = []
    for flags, release in ((0, False), (_lib.HULK_INGEST_HOST_PARSER, False), (0, True)):
        if release:
            assert _lib.load().hulk_release_caches() == 0
        g = gpu().GpuSketcher(21, 9, 64, interval=4000)
        st = g.sketch_files([p1, p2], opts={"flags": flags, "block_bytes": 131072})
        g.finish()
        res.append((st["n_seqs"], st["total_len"], st["n_lines"], g.sketch(), g.counters()))
        g.close()
    assert res[0][0] == len(want) and res[0][1] == sum(len(s) for s in want)
    for r in res[1:]:
        assert r[:3] == res[0][:3] and r[4] == res[0][4]
        assert np.array_equal(r[3][0], res[0][3][0]) and np.array_equal(r[3][1], res[0][3][1])


def _fa_run(paths, flags, k=11, w=5, S=32, interval=0, block=131072):
    """(stats triple, sketch, counters) of one hulk_sketch_files(--fasta) run, or the error's (code, message)"""
    from hulk_amd._lib import HulkError
    g = gpu().GpuSketcher(k, w, S, interval=interval)
    try:
        st = g.sketch_files(paths, fasta=True, opts={"flags": flags, "block_bytes": block})
        g.finish()
        return (st["n_seqs"], st["total_len"], st["n_lines"]), g.sketch(), g.counters()
    except HulkError as e:
        return ("error", e.code, e.message)
    finally:
        g.close()


def _same(a, b):
    if a[0] == "error" or b[0] == "error":
        return a == b
    return a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])


def test_device_fasta_parser_equals_the_host_parser(tmp_path):
    """--fasta with the line pump on the device (hulk_fastq.hip k_fa_*, the default) against the host's parser threads
    (HULK_INGEST_HOST_PARSER) and the literal restatement of sketch.go:102-135: records that span many 128 KiB blocks, sequence
    lines in front of the first header (dropped), CR/LF, headers back to back (an empty record: the reference's error), a header as
    the last line, an unterminated last line, a second input that continues the first one's record, an EMPTY line that ends the
    parsing in the first block, a later one and right behind a header, a line of 64 KiB (bufio.Scanner's error) in front of and
    behind an empty line, and no header at all."""
    from hulk_amd import _lib
    from oracle import linepump
    rng = np.random.default_rng(2024)
    acgt = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)

    def record(name, L, width, eol=b"\n"):
        seq = bytes(acgt[rng.integers(0, len(acgt), size=L)])
        return b">" + name + eol + b"".join(seq[i:i + width] + eol for i in range(0, L, width))

    def write(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        return p

    HOST = _lib.HULK_INGEST_HOST_PARSER
    good = (b"ACGTACGT\nTTTT\n" + record(b"c1 first", 700_000, 60) + record(b"c2", 40, 60) + record(b"c4 crlf", 250_000, 70, b"\r\n") +
            record(b"c5", 333_333, 61) + record(b"c6", 15, 80))
    cases = {"good": [write("good.fa", good)],
             "unterminated": [write("unterminated.fa", good.rstrip(b"\n"))],
             "two inputs": [write("a.fa", good[:900_001]), write("b.fa", good[900_001:])],
             "empty record": [write("empty_record.fa", good + b">e1\n>e2\n" + record(b"c7", 100, 60))],
             "last header": [write("last_header.fa", good + b">last_header_without_sequence\n")],
             "no header": [write("no_header.fa", b"ACGT\n" * 1000)],
             "nothing": [write("nothing.fa", b"")],
             "long line": [write("long.fa", good[:300_000] + b"C" * 65_536 + b"\n" + good[300_000:])],
             "long tail": [write("long_tail.fa", good[:300_000] + b"C" * 200_000)]}
    for cut in (5, 12_345, 400_000, 711_700, len(good) // 2, len(good) - 40):
        at = good.index(b"\n", cut) + 1
        cases[f"stop {cut}"] = [write(f"stop_{cut}.fa", good[:at] + b"\n" + b"A" * 70_000 + b"\n" + good[at:])]
        cases[f"stop crlf {cut}"] = [write(f"stopcr_{cut}.fa", good[:at] + b"\r\n" + good[at:])]
    seen_ok = seen_err = 0
    for name, paths in cases.items():
        dev = _fa_run(paths, 0)
        host = _fa_run(paths, HOST)
        assert _same(dev, host), (name, dev[0], host[0], dev[1:] if dev[0] == "error" else "", host[1:] if host[0] == "error" else "")
        for block in (131072 + 12288, 1 << 20):                 # other block sizes: other places where lines and records are cut
            assert _same(_fa_run(paths, 0, block=block), host), (name, block)
        if dev[0] == "error":
            seen_err += 1
            continue
        seen_ok += 1
        try:
            want = linepump.sequences(paths, fasta=True)
        except linepump.PumpError:
            want = None
        if want is not None:
            assert dev[0][0] == len(want) and dev[0][1] == sum(len(x) for x in want), name
    assert seen_ok >= 10 and seen_err >= 5, (seen_ok, seen_err)
    # with intervals (a flush per 2 sequences) and the default block size
    p = cases["good"]
    assert _same(_fa_run(p, 0, interval=2, block=0), _fa_run(p, HOST, interval=2, block=0))


def test_device_fasta_parser_batches_and_many_records(tmp_path):
    """The two ways a batch ends before the stream does: 64 MB of complete records (the record in progress then moves to the other
    accumulation buffer), and record offsets running short (600 k records of 20 bases through 128 KiB blocks).  Device parser =
    host parser = the same sequences through hulk_add_reads."""
    from hulk_amd import _lib, synth
    HOST = _lib.HULK_INGEST_HOST_PARSER
    # (a) 30 records of 5 Mbases, 80 per line: 150 MB of sequence, two hand-overs before the end
    p = str(tmp_path / "big.fa")
    n, L = 30, 5_000_000
    with open(p, "wb") as fh:
        for i in range(n):
            seq = synth.reads_numpy(1000 + i, 1, L)[0][:L].tobytes()
            fh.write(b">chr%d\n" % i + b"\n".join(seq[j:j + 80] for j in range(0, L, 80)) + b"\n")
    dev = _fa_run([p], 0, k=21, w=9, S=64, block=0)
    host = _fa_run([p], HOST, k=21, w=9, S=64, block=0)
    assert dev[0] == (n, n * L, n * (1 + (L + 79) // 80)) and _same(dev, host), (dev[0], host[0])
    # (b) many short records
    q = str(tmp_path / "many.fa")
    m = 600_000
    bases = synth.reads_numpy(7, m, 20)[0][:m * 20].reshape(m, 20)
    rows = np.empty((m, 24), dtype=np.uint8)
    rows[:, 0] = ord(">"); rows[:, 1] = ord("r"); rows[:, 2] = ord("\n"); rows[:, 3:23] = bases; rows[:, 23] = ord("\n")
    open(q, "wb").write(rows.tobytes())
    dev = _fa_run([q], 0)
    host = _fa_run([q], HOST)
    assert dev[0] == (m, m * 20, 2 * m) and _same(dev, host), (dev[0], host[0])
    # (c) one sequence longer than an accumulation buffer (192 MB): the buffer grows under the record in progress; a short one behind it
    r = str(tmp_path / "chromosome.fa")
    Lc = 230_000_000
    seq = synth.reads_numpy(4242, 1, Lc)[0][:Lc]
    rows = Lc // 100
    body = np.full((rows, 101), ord("\n"), dtype=np.uint8)
    body[:, :100] = seq.reshape(rows, 100)
    with open(r, "wb") as fh:
        fh.write(b">chr1\n"); fh.write(body.tobytes()); fh.write(b">tail\n" + b"ACGT" * 25 + b"\n")
    del body, seq
    dev = _fa_run([r], 0, k=21, w=9, S=64, block=0)
    host = _fa_run([r], HOST, k=21, w=9, S=64, block=0)
    assert dev[0] == (2, Lc + 100, rows + 3) and _same(dev, host), (dev[0], host[0])


@pytest.mark.parametrize("fasta", [False, True])
def test_stdin_is_sketched_like_the_file(tmp_path, fasta):
    """`hulk sketch` without -f reads STDIN (cmd/sketch.go:98-110; pipeline/sketch.go:45-52): the same bytes through a pipe — short
    reads at a time, no seeking — and as a file argument give the same sketch file, FASTQ and --fasta (both on the device parsers)."""
    import json
    from hulk_amd import synth
    n, L = (4000, 150) if not fasta else (12, 40_000)
    bases, _ = synth.reads_numpy(77, n, L)
    raw = bases[:n * L].tobytes()
    if fasta:
        data = b"".join(b">c%d\n" % i + b"\n".join(raw[i * L + j:i * L + min(j + 70, L)] for j in range(0, L, 70)) + b"\n" for i in range(n))
    else:
        data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, raw[i * L:(i + 1) * L], b"I" * L) for i in range(n))
    p = str(tmp_path / ("in.fa" if fasta else "in.fq"))
    open(p, "wb").write(data)
    common = [sys.executable, "-m", "hulk_amd", "sketch", "-k", "15", "-w", "5", "-s", "24", "-i", "500" if not fasta else "3"] + (["--fasta"] if fasta else [])
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    a = subprocess.run(common + ["-f", p, "-o", str(tmp_path / "from_file")], capture_output=True, env=env, timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    with open(p, "rb") as fh:
        b = subprocess.run(common + ["-o", str(tmp_path / "from_stdin")], stdin=fh, capture_output=True, env=env, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    # a pipe, written a little at a time
    c = subprocess.Popen(common + ["-o", str(tmp_path / "from_pipe")], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    for at in range(0, len(data), 100_003):
        c.stdin.write(data[at:at + 100_003]); c.stdin.flush()
    c.stdin.close()
    assert c.wait(timeout=600) == 0, c.stderr.read()[-2000:]

    def sketch_of(name):
        d = json.load(open(tmp_path / (name + ".json")))
        sig = d["signatures"][0]["Sketch"]
        return sig["mins"], sig["weights"], sig["md5sum"]
    f0 = sketch_of("from_file")
    assert f0 == sketch_of("from_stdin") == sketch_of("from_pipe")
    assert len(f0[0]) == 24
