"""GPU parity: libhulkhip (through the C ABI) vs the CPU oracle on the same inputs.
Bar: histogram / counters / count-min counters / `mins` bit-exact; `weights` within 1e-9
relative (north-star tolerance is 1e-5; fp64 literal re-evaluation does far better)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import pack_reads
from oracle import pyorc

pytestmark = pytest.mark.gpu

WEIGHT_RTOL = 1e-9
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu():
    import hulk_amd
    return hulk_amd


def random_reads(rng, n, length, alphabet=b"ACGT"):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    if np.isscalar(length):
        lens = np.full(n, length)
    else:
        lens = rng.integers(length[0], length[1] + 1, size=n)
    return [bytes(a[rng.integers(0, len(a), size=l)]) for l in lens]


def run_both(seqs, k, w, S, interval=0, num_bins=0, batches=1, decay=1.0, batch=0):
    o = pyorc.Sketcher(k, w, S, num_bins, decay, interval)
    g = gpu().GpuSketcher(k, w, S, interval, decay, num_bins, batch=batch)
    bases, offsets = pack_reads(seqs)
    o.add_reads(bases, offsets)
    # feed the GPU in several host batches to exercise interval splitting across calls
    n = len(seqs)
    cuts = np.linspace(0, n, batches + 1).astype(int)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            g.add_reads(bases, offsets[a:b + 1])
    return o, g


def assert_same_sketch(o, g):
    om, ow = o.sketch()
    gm, gw = g.sketch()
    assert np.array_equal(om, gm), f"{(om != gm).sum()} of {len(om)} mins differ"
    assert np.allclose(gw, ow, rtol=WEIGHT_RTOL, atol=0), np.max(np.abs(gw - ow) / np.abs(ow))


def test_histogram_fixture_bit_exact(fq_reads):
    """The reference's CI input, k=21 and k=31: spectrum identical to the oracle (and to the
    independently derived SHA-256 recorded in SURVEY.md App. C)."""
    sha = {21: "d4e4bf949482bdf426dd821c4bbe924ba9e395de5e4e9accc736bce888049de1",
           31: "732d2e405086edec32fb31204ee01ad8a83856669461f7b2ae05a11866ff6bf0"}
    for k in (21, 31):
        o, g = run_both(fq_reads, k, 9, 4)
        gh = g.histogram()
        assert np.array_equal(gh, o.histogram().astype(np.uint32))
        assert hashlib.sha256(gh.astype("<u4").tobytes()).hexdigest() == sha[k]
        oc, gc = o.counters(), g.counters()
        for key in ("n_reads", "n_minimizers", "total_len"):
            assert oc[key] == gc[key], key
        g.close(); o.close()


def test_sketch_fixture_c1(fq_reads):
    """BASELINE config C1: the test file, k=21, sketchSize=256, no interval."""
    o, g = run_both(fq_reads, 21, 9, 256)
    o.finish(); g.finish()
    assert_same_sketch(o, g)
    assert np.array_equal(g.cms(), o.cms())
    g.close(); o.close()


@pytest.mark.parametrize("k,w,S,n,L,interval", [
    (11, 5, 32, 3000, 150, 0),
    (11, 5, 32, 3000, 150, 500),        # 6 flushes: CMS carries across intervals
    (15, 9, 64, 4000, (40, 200), 1000),  # ragged lengths
    (21, 9, 16, 5000, 150, 2500),
    (7, 3, 8, 600, 64, 100),
    (31, 9, 8, 12000, 150, 0),          # k=31: the <<8 wraps
    # v_min_f64 path (k <= 27): values up to 2^62 are normal doubles (at k = 21 all of them are denormal
    # patterns); k = 28 is the first k that must use the integer compares again.  Enough reads for the 1 % rule.
    (25, 9, 3, 9000, (100, 150), 0), (27, 9, 2, 14000, (90, 151), 0), (28, 9, 2, 16000, (90, 151), 0),
    (23, 5, 4, 6000, (60, 120), 2500),
])
def test_random_reads(k, w, S, n, L, interval):
    rng = np.random.default_rng(k * 1000 + w)
    seqs = random_reads(rng, n, L)
    o, g = run_both(seqs, k, w, S, interval, batches=3)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    o.finish(); g.finish()
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    assert np.array_equal(g.cms(), o.cms())
    assert_same_sketch(o, g)
    g.close(); o.close()


def test_n_lowercase_and_min_length():
    """The reference does not special-case N (minimizer.go:118-122), keeps r unmasked, folds case,
    and accepts reads of exactly w+k-1 bases."""
    rng = np.random.default_rng(7)
    k, w = 11, 5
    seqs = random_reads(rng, 1500, (w + k - 1, 120), b"ACGTacgtNnUu")
    seqs += random_reads(rng, 500, w + k - 1)
    seqs += [b"N" * 40, b"ACGT" * 10, b"A" * 50, b"\x00\x01\x02\x03" * 8]
    o, g = run_both(seqs, k, w, 8)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    g.close(); o.close()


@pytest.mark.parametrize("k,w,L", [(21, 9, 150), (21, 9, 280), (31, 9, 150), (17, 5, 100), (12, 3, 60), (21, 12, 150)])
def test_every_byte_of_the_nt4_table(k, w, L):
    """seq_nt4_table (minimizer.go:23-40) has four kinds of bytes: ACGT in both cases, U / u (the code of T), the raw bytes
    0..3 (their own value) and everything else — N, the IUPAC letters, punctuation, bytes above 127 — code 4, which the
    recurrence does not special-case (minimizer.go:118-122).  The short-read kernel takes all of them but the raw 0..3 (those
    reads go to the generic kernel); (21, 12): an instance without the code-4 variant, where such reads are deferred as before.
    One foreign byte at every position of a read, every byte value once, several per read, next to an N, at both ends, in
    the first bases of the NEXT read: spectrum, minimizer count and sketch against the oracle."""
    rng = np.random.default_rng(77 + k + 100 * w + L)
    iupac = b"RYKMSWBDHVXrykmswbdhvx.-*Uu"
    seqs = []
    for p in range(L):                                   # one foreign byte at every position
        b = bytearray(random_reads(rng, 1, L)[0]); b[p] = iupac[p % len(iupac)]; seqs.append(bytes(b))
    for v in range(256):                                 # every byte value once, somewhere
        b = bytearray(random_reads(rng, 1, L)[0]); b[int(rng.integers(0, L))] = v; seqs.append(bytes(b))
    allb = bytes(range(256))
    for _ in range(300):                                 # several per read, any byte, N among them
        b = bytearray(random_reads(rng, 1, L)[0])
        for q in rng.integers(0, L, size=int(rng.integers(1, 7))):
            b[q] = allb[int(rng.integers(4, 256))] if rng.random() < 0.8 else ord("N")
        seqs.append(bytes(b))
    base = random_reads(rng, 1, L)[0]
    seqs += [b"R" + base[1:], base[:-1] + b"Y", b"U" + base[1:-1] + b"u", b"R" * L, b"U" * L, b"RN" + base[2:], b"\x03" + base[1:],
             base[:-1] + b"\x00", b"\x80" * L, base]
    seqs += random_reads(rng, 300, (w + k - 1, L), b"ACGTUuRYn")       # ragged
    seqs += random_reads(rng, 400, L)                    # clean reads in between: waves with and without a flag
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    o, g = run_both(seqs, k, w, 8, batches=3)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    o.finish(); g.finish()
    assert_same_sketch(o, g)
    g.close(); o.close()


@pytest.mark.parametrize("k,w,long_reads", [(21, 9, False), (21, 4, False), (21, 9, True), (21, 4, True),
                                            (31, 9, False), (31, 9, True), (17, 5, False), (24, 9, False), (12, 3, True)])
def test_fast_kernel_instance_with_n_bases(k, w, long_reads):
    """Reads with `N` stay in the short-read kernel's k = 21 instance (code 4 in closed form: minimizer.go:118-122 does not
    special-case it — bit 2 of the code spills into the neighbouring base's pair of f, r keeps bit 2k; k = 21: closed form on
    the one-window extraction, other k: the literal recurrence on the rolling k-mers): one N at every
    position of a 150 bp read, lower-case n, runs of N, N at both ends, several N per read, an N in the first bases of the
    NEXT read (the kernel stages past a read's end), reads of the minimum length, and reads that carry an N next to a byte
    the kernel does not take (those still go to the generic kernel); long_reads: a call whose reads take two 16-lane groups
    each (the second group's first k-mer needs the flag of a base its partner staged) — spectrum, minimizer count and
    sketch against the oracle."""
    rng = np.random.default_rng(2100 + w + 100 * k)
    L = 150
    base = random_reads(rng, 1, L)[0]
    seqs = []
    for p in range(L):                                   # one N at every position
        b = bytearray(random_reads(rng, 1, L)[0]); b[p] = ord("N" if p % 3 else "n"); seqs.append(bytes(b))
    for _ in range(400):                                 # several N, runs of N
        b = bytearray(random_reads(rng, 1, L)[0])
        for q in rng.integers(0, L, size=int(rng.integers(1, 6))):
            b[q] = ord("N")
        if rng.random() < 0.3:
            a = int(rng.integers(0, L - 12)); b[a:a + int(rng.integers(2, 12))] = b"N" * 11
        seqs.append(bytes(b[:L]))
    seqs += [b"N" + base[1:], base[:-1] + b"N", b"N" + base[1:-1] + b"N", b"N" * L, base, b"NN" + base[2:]]
    seqs += random_reads(rng, 300, (w + k - 1, 80), b"ACGTN")          # short and ragged, 20 % N
    seqs += random_reads(rng, 300, L, b"ACGTacgtNn")
    for _ in range(100):                                 # an N and a byte outside ACGTN in the same read
        b = bytearray(random_reads(rng, 1, L)[0]); b[int(rng.integers(0, L))] = ord("N"); b[int(rng.integers(0, L))] = ord("R"); seqs.append(bytes(b))
    seqs += random_reads(rng, 500, L)                    # clean reads in between: waves with and without an N
    if long_reads:                                       # reads of two 16-lane groups (250-300 bp): an N at every 3rd position,
        L2 = 16 * w + 100 if w == 9 else 16 * w + 30     # around the seam between the groups (position 16w - (w-1)) densely
        seqs = seqs[:600]
        seam = 16 * w - (w - 1)
        for p in list(range(0, L2, 3)) + list(range(seam - 3, seam + k + 3)):
            b = bytearray(random_reads(rng, 1, L2)[0]); b[p] = ord("N"); seqs.append(bytes(b))
        for _ in range(200):
            b = bytearray(random_reads(rng, 1, int(rng.integers(16 * w + k, L2 + 1)))[0])
            for q in rng.integers(0, len(b), size=int(rng.integers(1, 5))):
                b[q] = ord("N")
            seqs.append(bytes(b))
        seqs += random_reads(rng, 200, L2)
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    o, g = run_both(seqs, k, w, 8, batches=3)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    o.finish(); g.finish()
    assert_same_sketch(o, g)
    g.close(); o.close()


def test_even_k_symmetric_kmers_skipped():
    """even k: palindromic k-mers (f == r) are skipped (minimizer.go:145)."""
    rng = np.random.default_rng(3)
    seqs = random_reads(rng, 800, 90, b"AT") + [b"ATATATATATATATATATATATAT", b"ACGTACGTACGTACGTACGT"]
    o, g = run_both(seqs, 6, 4, 4, num_bins=997)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    g.close(); o.close()


def test_duplicate_minimizers_within_read():
    """per-read set semantics: a repeated k-mer counts once per read (minimizer.go:189-198)."""
    unit = b"ACGGTCATTGCAGTACCGTTAGC"
    seqs = [unit * 6, unit * 3 + b"TTTT" + unit * 3] * 50
    o, g = run_both(seqs, 9, 4, 4, num_bins=5000)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    g.close(); o.close()


def test_errors_match_reference_text():
    h = gpu()
    with pytest.raises(h.HulkError, match="histosketching only supports k <= 31"):
        h.GpuSketcher(32, 9, 8)
    with pytest.raises(h.HulkError, match="w must be: 0 < w < 257"):
        h.GpuSketcher(21, 300, 8)
    with pytest.raises(h.HulkError, match="histogram must have at least 2 bins"):
        h.GpuSketcher(1, 1, 8)
    g = h.GpuSketcher(21, 9, 8)
    with pytest.raises(h.HulkError, match="sequence length must be >= w \\+ k - 1"):
        g.add_seq(b"ACGT" * 7)          # 28 < 29
    g.close()
    # < 1 % of bins used -> "not used yet" (kmerspectrum.go:94-96), surfaced at finish
    g = h.GpuSketcher(21, 9, 8)
    g.add_seq(b"ACGTTGCATGCATGCAAAGTCGATCGATCGGGCTAGCTAGCTAGCTTTGAC")
    with pytest.raises(h.HulkError, match="not used yet"):
        g.finish()
    g.close()
    g = h.GpuSketcher(21, 9, 8)
    with pytest.raises(h.HulkError, match="no sequences received"):
        g.finish()
    g.close()


def test_cws_tables_match_oracle():
    g = gpu().GpuSketcher(9, 4, 6)
    r, c, b = g.cws_tables()
    orr, oc, ob = pyorc.cws_tables(6, 9 ** 4)
    for a, e in ((r, orr), (c, oc), (b, ob)):
        assert np.allclose(a, e, rtol=1e-13, atol=0)
    g.close()


def test_histogram_hook_sparse_and_dense():
    """AddElement parity driven directly by histograms, incl. very sparse ones where some slots
    keep positive minima (exclusion of zero bins must be exact)."""
    rng = np.random.default_rng(11)
    k, S = 7, 24
    B = k ** 4
    o = pyorc.Sketcher(k, 3, S); g = gpu().GpuSketcher(k, 3, S)
    for dens in (0.02, 0.5, 1.0, 0.011):
        hist = (rng.random(B) < dens) * rng.integers(1, 40, size=B)
        hist = hist.astype(np.uint32)
        o.add_histogram(hist); g.add_histogram(hist)
        o.flush(); g.flush()
        assert_same_sketch(o, g)
    g.close(); o.close()


def test_reciprocal_exhaustive():
    """The jump hash's 2^31/r uses a Newton reciprocal instead of the IEEE division sequence:
    checked on the device against IEEE division for every r in [1, 2^31]."""
    g = gpu().GpuSketcher(9, 4, 4)
    assert g.selftest_reciprocal() == 0
    g.close()


def test_jump_hash_many_keys_large_bins():
    """k=31 has 923,521 bins (20 jump iterations): spectrum of 30k random reads bit-exact."""
    rng = np.random.default_rng(99)
    seqs = random_reads(rng, 30000, 150)
    o, g = run_both(seqs, 31, 9, 2)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    g.close(); o.close()


@pytest.mark.parametrize("batch", [1, 3, 16])
def test_interval_batches_ring_wraparound(batch):
    """Intervals are flushed in batches of hulk_params.batch spectra held in a ring; uneven host calls leave
    partial intervals pending across calls and wrap the ring many times.  Must equal the oracle's
    flush-every-interval result bit for bit."""
    rng = np.random.default_rng(batch)
    k, w, S, interval = 11, 5, 16, 50
    seqs = random_reads(rng, 2113, (60, 150), b"ACGTN" if batch == 3 else b"ACGT")
    o = pyorc.Sketcher(k, w, S, 0, 1.0, interval)
    g = gpu().GpuSketcher(k, w, S, interval, batch=batch)
    assert g.batch_size == batch
    bases, offsets = pack_reads(seqs)
    o.add_reads(bases, offsets)
    cuts = [0, 7, 50, 51, 420, 1000, 1777, 2113]
    for a, b in zip(cuts[:-1], cuts[1:]):
        g.add_reads(bases, offsets[a:b + 1])
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))   # the pending partial interval
    o.finish(); g.finish()
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    assert np.array_equal(g.cms(), o.cms())
    assert_same_sketch(o, g)
    g.close(); o.close()


@pytest.mark.parametrize("lanes,batch,alph", [(2, 1, b"ACGT"), (2, 3, b"ACGTN"), (2, 8, b"ACGT"), (1, 3, b"ACGTN"), (2, 16, b"ACGTNacgt")])
def test_batches_on_alternating_work_lanes(lanes, batch, alph):
    """Consecutive batches are binned on two alternating work streams (one lane per spectrum ring, each with its own
    minimizer list; hulk_flush.hip, lane_stream) and are not ordered against each other.  Uneven host calls (partial
    intervals pending across calls keep a ring — and its lane — over several calls), reads with N (each lane's
    deferred-read list), reads of 250-400 bases in between (two groups per read / the generic kernel on the ring's lane):
    spectrum, counters, count-min and sketch equal the oracle's."""
    rng = np.random.default_rng(1000 + 10 * lanes + batch)
    k, w, S, interval = 15, 9, 24, 500
    seqs = random_reads(rng, 14_000, (40, 150), alph)      # (min length w + k - 1 = 23)
    for i in range(5000, 5400):
        seqs[i] = random_reads(rng, 1, 250 + (i % 3) * 75, alph)[0]
    o = pyorc.Sketcher(k, w, S, 0, 1.0, interval)
    g = gpu().GpuSketcher(k, w, S, interval, batch=batch, work_lanes=lanes)
    bases, offsets = pack_reads(seqs)
    o.add_reads(bases, offsets)
    cuts = [0, 3999, 4000, 9001, 9300, 14_000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        g.add_reads(bases, offsets[a:b + 1])
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))   # (14,000 = 28 whole intervals: empty)
    o.finish(); g.finish()
    oc, gc = o.counters(), g.counters()
    for key in ("n_reads", "n_minimizers", "total_len"):
        assert oc[key] == gc[key], key
    assert np.array_equal(g.cms(), o.cms())
    assert_same_sketch(o, g)
    g.close(); o.close()


def test_bin_then_flush_batch_equals_interval_rule():
    """The multi-GPU entry points (bin_reads_device + flush_batch) on one GPU = the interval rule."""
    import torch
    from hulk_amd import synth
    k, w, S, interval, T = 13, 7, 12, 300, 5
    g = gpu().GpuSketcher(k, w, S, 0, batch=T)
    assert g.batch_size == T
    o = pyorc.Sketcher(k, w, S, 0, 1.0, interval)
    for step in range(3):
        b, off = synth.reads_torch(step * interval * T, interval * T, 120)
        torch.cuda.synchronize()        # generated on torch's stream; the context runs on its own
        g.bin_reads_device(b.data_ptr(), off.data_ptr(), interval * T, 120, b.numel(), interval)
        g.flush_batch(T)
        hb, ho = synth.reads_numpy(step * interval * T, interval * T, 120)
        o.add_reads(hb, ho)
    torch.cuda.synchronize()
    o.finish(); g.finish()
    assert_same_sketch(o, g)
    assert np.array_equal(g.cms(), o.cms())
    g.close(); o.close()


DRIFT_RTOL = 1e-7   # the count-min decay is evaluated in closed form (w^gap) instead of step by step


@pytest.mark.parametrize("decay,k,w,S,n,L,interval", [
    (0.02, 9, 4, 24, 3000, 120, 500),
    (0.5, 9, 4, 24, 3000, 120, 500),
    (0.002, 11, 5, 16, 4000, 150, 1000),
    (0.0, 9, 4, 16, 2000, 120, 500),      # drift on, scaling off, decayWeight == 0 (w/0 semantics)
    (0.9999, 7, 3, 8, 1500, 100, 0),
])
def test_concept_drift(decay, k, w, S, n, L, interval):
    """decay_ratio != 1: uniform scaling of the count-min counters (countmin.go:141-147) and the
    `A < w/decayWeight` update (histosketch.go:139-153), BASELINE config C3's mode."""
    rng = np.random.default_rng(int(decay * 1e4) + k)
    seqs = random_reads(rng, n, L)
    o, g = run_both(seqs, k, w, S, interval, batches=2, decay=decay)
    o.finish(); g.finish()
    om, ow = o.sketch(); gm, gw = g.sketch()
    assert np.array_equal(om, gm), f"{(om != gm).sum()} of {len(om)} mins differ"
    assert np.allclose(gw, ow, rtol=DRIFT_RTOL, atol=0)
    oc, gc = o.cms(), g.cms()
    assert np.allclose(gc, oc, rtol=1e-9, atol=1e-300)
    g.close(); o.close()


def test_cli_c1_fixture(tmp_path, fq_reads):
    """BASELINE config C1 end to end: `hulk sketch -f test-reads-small.fq.gz -k 21 --sketchSize 256`
    through the flag-compatible front-end; JSON fields vs the oracle."""
    import json, os
    from conftest import GOLDEN
    from hulk_amd.__main__ import main
    from hulk_amd.sketchio import load_hulk_data
    out = str(tmp_path / "sk")
    fq = os.path.join(GOLDEN, "test-reads-small.fq.gz")
    assert main(["sketch", "-f", fq, "-k", "21", "--sketchSize", "256", "-o", out]) == 0
    doc = json.load(open(out + ".json"))
    assert doc["filename"] == fq + "," and doc["class"] == "hulk_sketch" and doc["version"] == "1.0.0"
    sk = doc["signatures"][0]["Sketch"]
    assert sk["ksize"] == 21 and sk["num"] == 256 and sk["num_histogram_bins"] == 194481 and sk["concept_drift"] is False
    o = pyorc.Sketcher(21, 9, 256)
    for r in fq_reads:
        o.add_read(r)
    o.finish()
    om, ow = o.sketch()
    assert sk["mins"] == om.tolist()
    assert np.allclose(np.array(sk["weights"]), ow, rtol=1e-5, atol=0)          # north-star tolerance on the JSON
    assert np.allclose(np.array(sk["weights"]), ow, rtol=WEIGHT_RTOL, atol=0)
    load_hulk_data(out + ".json")                                              # class/version/MD5 checks
    # error path: a read shorter than w+k-1 is fatal with the reference's message
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r\nACGTACGT\n+\nIIIIIIII\n")
    assert main(["sketch", "-f", str(bad), "-o", out]) == 1


@pytest.mark.parametrize("k,w,lens,n", [
    (15, 9, (300, 900), 600),        # generic kernel, 1024-position configuration
    (15, 9, (1500, 4000), 200),      # generic kernel, 4096-position configuration
    (21, 16, (150, 150), 2000),      # fast kernel, widest block (WM = 16)
    (21, 1, (30, 36), 3000),         # w = 1: every k-mer is its own minimizer; <= 16 positions on the fast path
    (21, 20, (150, 250), 1500),      # w > 16: generic kernel only
    (12, 4, (40, 300), 2500),        # mixed: short reads on the fast path, long ones deferred to the generic kernel
    (9, 0, (20, 60), 1000),          # w = 0 is accepted by the reference (windowIndex = i + 1)
])
def test_read_length_and_window_configurations(k, w, lens, n):
    rng = np.random.default_rng(k * 100 + w)
    seqs = random_reads(rng, n, lens)
    o, g = run_both(seqs, k, w, 4, num_bins=200003)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    g.close(); o.close()


def test_long_sequences_fasta_style():
    """Sequences beyond the generic kernel's 4096 positions (FASTA contigs) take the long-sequence
    path: per-sequence set in an HBM hash table.  Mixed with short and medium reads, with N."""
    rng = np.random.default_rng(12)
    seqs = random_reads(rng, 3, (20000, 40000), b"ACGTN" * 40 + b"N")[0:3]
    seqs += random_reads(rng, 300, (60, 150)) + random_reads(rng, 20, (2000, 4000))
    seqs += [seqs[0][:9000] * 3]                         # a long repetitive one: duplicates inside one sequence
    rng.shuffle(seqs)
    o, g = run_both(seqs, 15, 9, 4, num_bins=300007, batches=2)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    g.close(); o.close()


@pytest.mark.parametrize("k,w", [(21, 9), (5, 1), (31, 16), (15, 40), (11, 255), (21, 256), (8, 8)])
def test_long_tile_kernel_at_its_tile_borders(k, w):
    """k_long_tile (round 6): a workgroup owns 2048 consecutive k-mer positions of a sequence, the first w rounded up to 8 of them
    context.  Sequences whose number of positions sits on and around multiples of the tile's reporting span, for every shape of w
    (1: every position reports; 8: w itself a multiple of 8; 255 / 256: the largest windows, 256 context positions), code-4 bytes
    right at the tile borders (the N of base p reaches the k-mers of p - 1 and p + 1: minimizer.go:118-122), a sequence made of one
    repeated unit (the per-sequence set sees every value again in every tile), lower case — spectrum and minimizer count against the
    oracle; and the profiling build's two-pass kernels (HULK_LONG_TWO_PASS) give the same."""
    rng = np.random.default_rng(100 * k + w)
    H = (max(w, 1) + 7) // 8 * 8
    TP = 2048 - H
    seqs = []
    for npos in (1025, TP - 1, TP, TP + 1, 2 * TP, 2 * TP + 1, 3 * TP - 1, 3 * TP + H):
        L = npos + k - 1
        if L < w + k - 1:
            continue
        s = bytearray(random_reads(rng, 1, L, b"ACGTacgt")[0])
        for border in (TP, 2 * TP):                                # code-4 bytes where one tile ends and the next begins
            for at in (border - 1, border, border + k - 1, border + k):
                if 0 <= at < L and rng.random() < 0.5:
                    s[at] = ord("N")
        seqs.append(bytes(s))
    unit = random_reads(rng, 1, 700)[0]
    seqs.append(unit * 9)                                          # 6300 bases: the same ~140 minimizers in every tile
    seqs += random_reads(rng, 40, (w + k - 1, 400))                # short reads beside them (other kernels, same spectrum)
    B = min(k ** 4, 300007) if k ** 4 > 2 else 16
    o, g = run_both(seqs, k, w, 4, num_bins=B if k < 9 else 0, batches=2)
    gh = g.histogram()
    assert np.array_equal(gh, o.histogram().astype(np.uint32)), f"{int((gh != o.histogram()).sum())} bins differ"
    n_min = o.counters()["n_minimizers"]
    assert n_min == g.counters()["n_minimizers"]
    g.close(); o.close()
    # the two-pass comparator (profiling build)
    import subprocess
    import sys
    import tempfile
    if not os.path.exists(os.path.join(ROOT, "hulk_amd", "csrc", "libhulkhip_exp.so")):
        return
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "seqs.npy"), np.array(seqs, dtype=object), allow_pickle=True)
        code = (
            "import sys, hashlib, numpy as np\n"
            f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
            "import torch, hulk_amd\n"
            "from conftest import pack_reads\n"
            "seqs = list(np.load(sys.argv[1], allow_pickle=True))\n"
            f"g = hulk_amd.GpuSketcher({k}, {w}, 4, num_bins={B if k < 9 else 0})\n"
            "g.add_reads(*pack_reads(seqs))\n"
            "print(hashlib.md5(g.histogram().tobytes()).hexdigest(), g.counters()['n_minimizers'])\n"
            "g.close()\n")
        r = subprocess.run([sys.executable, "-c", code, os.path.join(td, "seqs.npy")], env=dict(os.environ, HULK_LIB="exp", HULK_LONG_TWO_PASS="1"),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        md5, nmin = r.stdout.split()[-2:]
        import hashlib
        assert md5 == hashlib.md5(gh.tobytes()).hexdigest() and int(nmin) == n_min


def test_repetitive_reads_fall_back_to_generic_kernel():
    """> 64 run starts in one read (low-complexity / tandem repeats) leave the fast path."""
    rng = np.random.default_rng(4)
    unit = bytes(rng.choice(list(b"ACGT"), size=7))
    seqs = [unit * 21] * 40 + random_reads(rng, 500, 147) + [b"AC" * 70, b"A" * 150, b"ACG" * 50]
    o, g = run_both(seqs, 5, 2, 4, num_bins=997)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    g.close(); o.close()


@pytest.mark.parametrize("name,k,S,interval,decay", [("c1_fixture_k21_s256", 21, 256, 0, 1.0),
                                                     ("c1_fixture_k15_s64_drift", 15, 64, 250, 0.05)])
def test_against_committed_golden_sketches(name, k, S, interval, decay, fq_reads):
    """GPU sketch of the reference's fixture vs the committed golden JSON (tests/golden, produced by
    tools/make_golden.py from the CPU restatement): mins + MD5 exact, weights within the north-star 1e-5."""
    import os
    from conftest import GOLDEN
    from hulk_amd.sketchio import load_hulk_data, md5sum
    gold = load_hulk_data(os.path.join(GOLDEN, name + ".json")).signatures[0][1]
    g = gpu().GpuSketcher(k, 9, S, interval, decay)
    g.add_reads(*pack_reads(fq_reads))
    g.finish()
    hs = g.histosketch()
    assert np.array_equal(hs.mins, gold.mins) and md5sum(hs.mins) == gold.md5sum
    assert np.allclose(hs.weights, gold.weights, rtol=1e-5, atol=0)
    assert np.allclose(hs.weights, gold.weights, rtol=1e-7, atol=0)
    g.close()


def test_many_long_reads_grouped_launches():
    """Long reads (beyond the one-wave kernel's 4096 positions) go through the grouped long-sequence path:
    240 reads of 4.2-30 kb with N runs, lowercase and repeats, intervals crossing inside the batch,
    mixed with short and medium reads in the same call."""
    rng = np.random.default_rng(99)
    seqs = []
    for i in range(240):
        L = int(rng.integers(4200, 30000))
        s = bytearray(random_reads(rng, 1, L, b"ACGTacgt")[0])
        if i % 7 == 0:
            a = int(rng.integers(0, L - 300)); s[a:a + int(rng.integers(1, 250))] = b"N" * 1      # shortens: fine
        if i % 5 == 0:
            a = int(rng.integers(0, L // 2)); s[a + 500:a + 1500] = s[a:a + 1000]                    # internal repeat
        if i % 11 == 0:
            s = s[:5000] + b"N" * 40 + s[5000:]
        seqs.append(bytes(s))
        if i % 3 == 0:
            seqs.extend(random_reads(rng, 4, (60, 150)))
        if i % 17 == 0:
            seqs.extend(random_reads(rng, 1, (500, 3000)))
    o, g = run_both(seqs, 15, 9, 32, interval=37, batches=3)
    o.finish(); g.finish()
    oc, gc = o.counters(), g.counters()
    assert all(oc[key] == gc[key] for key in ("n_reads", "n_minimizers", "total_len"))
    assert np.array_equal(g.cms(), o.cms())
    assert_same_sketch(o, g)
    g.close(); o.close()


def test_scan_pruning_is_exact_and_effective():
    """The bound test of k_cws_scan must not change a single bit of the sketch, and after the first
    intervals of a stream it must skip most of the table (count-min estimates only grow)."""
    from hulk_amd import synth
    bases, offsets = synth.reads_numpy(0, 60000, 150)
    res = {}
    from hulk_amd import _lib
    for prune in (True, False, "tiles-only"):
        # False: neither bound; "tiles-only": per-tile bound only, no whole-batch bound
        flags = _lib.HULK_FLAG_NO_PRUNE if prune is False else _lib.HULK_FLAG_NO_SKIP if prune == "tiles-only" else 0
        g = gpu().GpuSketcher(15, 9, 96, interval=2000, batch=4, flags=flags)
        g.add_reads(bases, offsets)
        g.finish()
        res[prune] = (g.sketch(), g.scan_stats(), g.cms())
        g.close()
    (m1, w1), (v1, t1), c1 = res[True]
    (m0, w0), (v0, t0), c0 = res[False]
    assert np.array_equal(m1, m0) and np.array_equal(w1, w0) and np.array_equal(c1, c0)
    (m2, w2), (v2, t2), c2 = res["tiles-only"]
    assert np.array_equal(m2, m0) and np.array_equal(w2, w0) and np.array_equal(c2, c0) and v2 >= v1
    assert t1 == t0 and v0 == t0                       # unpruned: every covered tile is read
    assert v1 < 0.35 * t1, (v1, t1)                    # 30 intervals: only the first batches read everything
    o = pyorc.Sketcher(15, 9, 96, 0, 1.0, 2000)
    o.add_reads(bases, offsets)
    o.finish()
    assert np.array_equal(o.sketch()[0], m1) and np.allclose(o.sketch()[1], w1, rtol=WEIGHT_RTOL, atol=0)
    o.close()


def test_nibble_histogram_overflow_falls_back_to_exact_count():
    """k_nibble_hist counts with 4-bit counters; thousands of identical reads push single bins far past 15 per
    part, which must be detected (nib_over) and recounted exactly — per spectrum, so clean and overflowing
    intervals mix in one launch."""
    rng = np.random.default_rng(5)
    same = random_reads(rng, 1, 150)[0]
    seqs = random_reads(rng, 4000, 150)                      # interval 0: clean
    seqs += random_reads(rng, 1000, 150) + [same] * 3000      # interval 1: 3000 hits in ~17 bins (+ enough bins for the 1 % rule)
    seqs += [same] * 2000 + random_reads(rng, 2000, 150)      # interval 2: mixed, the copies first
    seqs += random_reads(rng, 4000, 150)                      # interval 3: clean again
    o, g = run_both(seqs, 15, 9, 24, interval=4000, batches=1)
    o.finish(); g.finish()
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    assert np.array_equal(g.cms(), o.cms())
    assert_same_sketch(o, g)
    g.close(); o.close()


@pytest.mark.parametrize("k,w,lens,n", [
    (21, 9, (250, 250), 4000),        # MiSeq 2x250: two 16-lane groups per read
    (21, 9, (300, 300), 4000),        # exactly the capacity: 280 positions
    (21, 9, (29, 300), 5000),         # everything from the minimum length up, in one launch
    (21, 9, (290, 330), 3000),        # some fit the pair (<= 300), the others go to the one-wave kernel
    (15, 5, (100, 169), 4000),        # w = 5: 80 / 156 positions
    (27, 16, (260, 300), 3000),       # 16w + k - 1 > 256: the pair mode must not be used
    (9, 4, (64, 135), 3000),
    (7, 3, (100, 100), 3000),         # dense: ~51 distinct minimizers per read, more than one group's 16w positions
    (5, 1, (21, 36), 3000),           # w = 1: every position is a minimizer
    (6, 2, (40, 66), 3000),           # even k: self-complementary k-mers are skipped
])
def test_two_groups_per_read(k, w, lens, n):
    """Reads beyond one group's 16w positions take two neighbouring groups (k_minimizer_fast<..., PAIR>): the
    second one re-derives w-1 positions of context, reports none of them, and shares the read's set."""
    rng = np.random.default_rng(k * 1000 + w)
    seqs = random_reads(rng, n, lens)
    for i in range(0, n, 97):                                  # a few N reads and repeats (deferred / deduplicated)
        s = bytearray(seqs[i]); s[len(s) // 2] = ord("N"); seqs[i] = bytes(s)
    for i in range(5, n, 101):
        s = seqs[i]; seqs[i] = (s[:60] * 6)[:len(s)]
    o, g = run_both(seqs, k, w, 8, interval=1500, batches=2)
    o.finish(); g.finish()
    oc, gc = o.counters(), g.counters()
    assert all(oc[key] == gc[key] for key in ("n_reads", "n_minimizers", "total_len"))
    assert np.array_equal(g.cms(), o.cms())
    assert_same_sketch(o, g)
    g.close(); o.close()


@pytest.mark.parametrize("batch", [1, 3, 16])
@pytest.mark.parametrize("decay", [0.02, 0.5])
def test_k31_concept_drift_against_oracle(decay, batch):
    """BASELINE config C3's mode at its k: k = 31 (923,521 bins, minimizer values use all 64 bits: integer minima,
    rolling k-mers) WITH concept drift — count-min uniform scaling (countmin.go:141-147) and the
    `A < w/decayWeight` update (histosketch.go:139-153) — against the oracle, for three interval-batch sizes."""
    rng = np.random.default_rng(31_000 + int(decay * 100))
    seqs = random_reads(rng, 30_000, 150)
    o, g = run_both(seqs, 31, 9, 4, interval=5_000, batches=3, decay=decay, batch=batch)
    assert g.batch_size == batch
    o.finish(); g.finish()
    om, ow = o.sketch(); gm, gw = g.sketch()
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    assert np.array_equal(om, gm), f"{(om != gm).sum()} of {len(om)} mins differ"
    assert np.allclose(gw, ow, rtol=DRIFT_RTOL, atol=0)
    assert np.allclose(g.cms(), o.cms(), rtol=1e-9, atol=1e-300)
    g.close(); o.close()


def test_concept_drift_is_reproducible_to_the_bit():
    """Drift mode (k = 31, decay 0.02: fp64 count-min counters with uniform scaling, countmin.go:141-147) gives the same
    `mins`, `weights` AND counters bit for bit in repeated runs: the segment sums of k_cmsd_segsum are formed in a fixed
    order (one wave per count-min row, ascending bins) — until round 3 fp64 LDS atomics of eight waves added in whatever
    order they arrived (~1e-13 between runs)."""
    from hulk_amd import synth
    bases, offsets = synth.reads_numpy(0, 60_000, 150)
    outs = []
    for _ in range(5):
        g = gpu().GpuSketcher(31, 9, 64, interval=20_000, decay_ratio=0.02)
        g.add_reads(bases, offsets)
        g.finish()
        m, w = g.sketch()
        outs.append((m, w, g.cms()))
        g.close()
    for m, w, c in outs[1:]:
        assert np.array_equal(m, outs[0][0])
        assert np.array_equal(w.view(np.uint64), outs[0][1].view(np.uint64))
        assert np.array_equal(c.view(np.uint64), outs[0][2].view(np.uint64))


def _numpy_histosketch(hist, r, c, b):
    """One flush of a histogram through count-min (countmin.go:103-147, no decay) and AddElement
    (histosketch.go:129-155) for ARBITRARY tables r, c, b [S][B] — a numpy restatement for the external-table test."""
    S, B = r.shape
    ctr = np.zeros((7, 2000))
    pos = np.array([[pyorc.jump((x + d * x) & 0xFFFFFFFFFFFFFFFF, 2000) for x in range(B)] for d in range(7)])
    mins = np.zeros(S, dtype=np.uint64); wts = np.full(S, np.finfo(np.float64).max)
    for x in np.nonzero(hist)[0]:
        f = np.inf
        for d in range(7):
            ctr[d, pos[d, x]] += float(hist[x]); f = min(f, ctr[d, pos[d, x]])
        A = c[:, x] / (np.exp(np.log(f) - b[:, x]) * np.exp(r[:, x]))
        upd = A < wts
        mins[upd] = x; wts[upd] = A[upd]
    return mins, wts


def test_external_cws_tables_are_what_the_sketch_follows():
    """hulk_set_cws_tables / HULK_CWS_EXTERNAL (include/hulk_hip.h) — the hook that takes r, c, b dumped by a real Go
    run of newCWS (histosketch.go:95-126; tools/go/dump_cws):
      * fed the oracle's tables it reproduces the default (device-generated) sketch bit for bit;
      * fed DIFFERENT tables (a slot permutation of the oracle's; tables drawn by numpy) the sketch follows the
        supplied ones and differs from the default;
      * HULK_FLAG_GAMMA_CPYTHON generates the same tables as the default (the squeeze constant is immaterial);
      * a context created EXTERNAL refuses to sketch before the tables are set."""
    h = gpu()
    from hulk_amd import _lib
    k, w, S, B = 11, 5, 24, 11 ** 4
    rng = np.random.default_rng(77)
    seqs = random_reads(rng, 4000, 150)
    bases, offsets = pack_reads(seqs)

    def gpu_sketch(tables=None, flags=0):
        g = h.GpuSketcher(k, w, S, 1000, cws_source=_lib.HULK_CWS_EXTERNAL if tables is not None else _lib.HULK_CWS_GO_COMPAT,
                          flags=flags)
        if tables is not None:
            g.set_cws_tables(*tables)
        g.add_reads(bases, offsets); g.finish()
        out = g.sketch(); g.close()
        return out

    o = pyorc.Sketcher(k, w, S, 0, 1.0, 1000)
    t0 = pyorc.cws_tables(S, B)
    o.add_reads(bases, offsets); o.finish()
    m0, w0 = o.sketch(); o.close()
    gm, gw = gpu_sketch()                                         # default: generated on the device
    assert np.array_equal(gm, m0) and np.allclose(gw, w0, rtol=WEIGHT_RTOL, atol=0)
    em, ew = gpu_sketch(t0)                                       # the same tables through the hook
    assert np.array_equal(em, gm) and np.allclose(ew, gw, rtol=1e-12, atol=0)
    fm, fw = gpu_sketch(flags=_lib.HULK_FLAG_GAMMA_CPYTHON)       # CPython's squeeze constant: same tables
    assert np.array_equal(fm, gm) and np.array_equal(fw, gw)
    perm = np.roll(np.arange(S), 5)                               # slot i gets the parameters of slot perm[i]
    pm, pw = gpu_sketch(tuple(np.ascontiguousarray(a[perm]) for a in t0))
    assert np.array_equal(pm, m0[perm]) and np.allclose(pw, w0[perm], rtol=WEIGHT_RTOL, atol=0)
    assert not np.array_equal(pm, m0)
    g = h.GpuSketcher(k, w, S, 1000, cws_source=_lib.HULK_CWS_EXTERNAL)
    with pytest.raises(h.HulkError, match="hulk_set_cws_tables"):
        g.add_reads(bases, offsets); g.finish()
    g.close()
    # tables that no generator of this repository produced: numpy's gamma / uniform, one histogram, numpy restatement
    k2, S2 = 7, 16
    B2 = k2 ** 4
    r = rng.gamma(2.0, 1.0, size=(S2, B2)); c = np.log(rng.gamma(2.0, 1.0, size=(S2, B2))); b = rng.random((S2, B2)) * r
    hist = ((rng.random(B2) < 0.6) * rng.integers(1, 30, size=B2)).astype(np.uint32)
    g = h.GpuSketcher(k2, 3, S2, cws_source=_lib.HULK_CWS_EXTERNAL)
    g.set_cws_tables(r, c, b)
    g.add_histogram(hist); g.flush()
    xm, xw = g.sketch(); g.close()
    nm, nw = _numpy_histosketch(hist, r, c, b)
    assert np.array_equal(xm, nm) and np.allclose(xw, nw, rtol=WEIGHT_RTOL, atol=0)
    d = h.GpuSketcher(k2, 3, S2)                                  # ... and the default tables give another sketch
    d.add_histogram(hist); d.flush()
    assert not np.array_equal(d.sketch()[0], xm)
    d.close()


def test_num_bins_limit():
    """The binning kernels pack (spectrum slot << 20 | bin): 2^20 bins is the largest spectrum (k^4 at k = 31 is
    923,521); one more is refused at hulk_create instead of aliasing bins silently.  Parity at the limit."""
    h = gpu()
    from hulk_amd import _lib
    with pytest.raises(h.HulkError, match="HULK_MAX_BINS"):
        h.GpuSketcher(15, 9, 2, num_bins=_lib.HULK_MAX_BINS + 1)
    rng = np.random.default_rng(20)
    seqs = random_reads(rng, 3000, 150)
    o, g = run_both(seqs, 15, 9, 2, interval=1500, num_bins=_lib.HULK_MAX_BINS)
    assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32))
    o.finish(); g.finish()
    assert_same_sketch(o, g)
    g.close(); o.close()


def test_cli_khf_kmv_stream_flags(tmp_path):
    """`hulk sketch --khf / --kmv / --stream` observable behaviour (cmd/sketch.go:56-59,75-86,121-122;
    pipeline/sketch.go:226-234; sketchio.go:56-75): --khf appends the never-fed KHF signature (all math.MaxUint64),
    --kmv is fatal after the reads were processed and leaves no JSON, --stream sends the log to <outFile>.log."""
    import json, os
    from conftest import GOLDEN
    from hulk_amd.__main__ import main
    from hulk_amd.sketchio import load_hulk_data
    fq = os.path.join(GOLDEN, "test-reads-small.fq.gz")
    out = str(tmp_path / "a")
    assert main(["sketch", "-f", fq, "-s", "16", "--khf", "-o", out]) == 0
    doc = json.load(open(out + ".json"))
    assert [s["Algorithm"] for s in doc["signatures"]] == ["histosketch", "khf"]
    assert doc["signatures"][1]["Sketch"] == {"ksize": 21, "md5sum": hashlib.md5(b"\xff" * 128).hexdigest(),
                                              "mins": [2 ** 64 - 1] * 16, "num": 16}
    load_hulk_data(out + ".json")
    plain = str(tmp_path / "p")
    assert main(["sketch", "-f", fq, "-s", "16", "-o", plain]) == 0
    assert json.load(open(plain + ".json"))["signatures"][0] == doc["signatures"][0]
    bad = str(tmp_path / "b")
    assert main(["sketch", "-f", fq, "-s", "16", "--kmv", "--khf", "-o", bad]) == 1
    assert not os.path.exists(bad + ".json")
    st = str(tmp_path / "logs" / "s")
    assert main(["sketch", "-f", fq, "-s", "16", "--stream", "-o", st]) == 0
    log = open(st + ".log").read()
    assert "\tstreaming: enabled" in log and "\tadding KHF sketch: false" in log and "written sketch to disk" in log
    assert json.load(open(st + ".json"))["signatures"][0] == doc["signatures"][0]


def test_import_order_does_not_matter():
    """hulk_amd first, torch afterwards, in a fresh process: both must see the GPU (one HIP runtime per process;
    hulk_amd/_lib.py maps torch's bundled runtime before libhulkhip.so binds to /opt/rocm's)."""
    import subprocess, sys, os
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, hulk_amd\n"
            "assert 'torch' not in sys.modules\n"
            "from hulk_amd import synth\n"
            "g = hulk_amd.GpuSketcher(15, 9, 8)\n"
            "b, o = synth.reads_numpy(0, 2000, 150)\n"
            "g.add_reads(b, o); g.finish(); m0 = g.sketch()[0]\n"
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "x = torch.arange(10, device='cuda').sum().item(); assert x == 45\n"
            "bt, ot = synth.reads_torch(0, 2000, 150); torch.cuda.synchronize()\n"
            "h = hulk_amd.GpuSketcher(15, 9, 8)\n"
            "h.add_reads_device(bt.data_ptr(), ot.data_ptr(), 2000, 150, bt.numel()); h.finish()\n"
            "assert np.array_equal(h.sketch()[0], m0)\n"
            "print('ok')\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-2000:]


def test_concept_drift_many_seeds_mins_exact():
    """Drift mode over many small streams: `mins` must equal the oracle's exactly every time (the closed-form decay and the
    re-associated segment sums only move `weights` in the last digits; a near-tie `A < w/decayWeight` flipping a slot would
    show here), and a repeated run reproduces the weights to 1e-12."""
    for seed in range(12):
        rng = np.random.default_rng(9000 + seed)
        decay = [0.02, 0.3, 0.002, 0.9][seed % 4]
        k = [9, 11, 13][seed % 3]
        seqs = random_reads(rng, 2500, (60, 180))
        o, g = run_both(seqs, k, 5, 20, interval=400, batches=2, decay=decay)
        o.finish(); g.finish()
        om, ow = o.sketch(); gm, gw = g.sketch()
        assert np.array_equal(om, gm), (seed, decay, k, int((om != gm).sum()))
        assert np.allclose(gw, ow, rtol=DRIFT_RTOL, atol=0)
        o2, g2 = run_both(seqs, k, 5, 20, interval=400, batches=1, decay=decay)
        g2.finish()
        gm2, gw2 = g2.sketch()
        assert np.array_equal(gm2, gm) and np.allclose(gw2, gw, rtol=1e-12, atol=0)
        for x in (o, g, o2, g2):
            x.close()


@pytest.mark.parametrize("S", [2048, 1001])
def test_large_and_ragged_sketch_sizes(S):
    """sketchSize 2048 is what BASELINE C5's sketches have; 1001 is not a multiple of the 8-slot scan groups (the last
    group's bound test, slot minima and resolve see 1 live row of 8).  k = 15, 4 intervals, against the oracle."""
    rng = np.random.default_rng(S)
    seqs = random_reads(rng, 20_000, 150)
    o, g = run_both(seqs, 15, 9, S, interval=5_000, batches=3)
    o.finish(); g.finish()
    assert_same_sketch(o, g)
    assert np.array_equal(g.cms(), o.cms())
    g.close(); o.close()
    if S == 1001:                                   # ... and with concept drift
        o, g = run_both(seqs[:8000], 15, 9, S, interval=2_000, batches=2, decay=0.05)
        o.finish(); g.finish()
        om, ow = o.sketch(); gm, gw = g.sketch()
        assert np.array_equal(om, gm) and np.allclose(gw, ow, rtol=DRIFT_RTOL, atol=0)
        g.close(); o.close()


def test_pair_sets_start_empty_low_complexity_reads():
    """Regression (found by tools/fuzz_parity.py, 12,000 cases): when two 16-lane groups share a read their per-read set spans
    BOTH groups' LDS tables; each group must empty its own at kernel start (for a while both emptied the first, and stale LDS
    in the second made some values look already seen — dropped minimizers, depending on what ran before).  Low-complexity
    reads over {A, C} (many repeats per read) in the two-groups shape (k = 5, w = 5, 150 bp), after other batches."""
    h = gpu()
    rng = np.random.default_rng(2298)
    for trial in range(6):
        k, w = 5, 5
        g = h.GpuSketcher(k, w, 1); o = pyorc.Sketcher(k, w, 1, 0, 1.0, 0)
        parts = [random_reads(rng, int(rng.integers(60, 300)), int(rng.integers(165, 250))),       # generic kernel
                 random_reads(rng, int(rng.integers(60, 300)), int(rng.integers(165, 250))),
                 random_reads(rng, int(rng.integers(20, 80)), 150, alphabet=b"AC") + [b"A" * 150, b"C" * 150, b"AC" * 75],
                 random_reads(rng, 7, 150, alphabet=b"AC")]
        for p in parts:
            b, off = pack_reads(p)
            g.add_reads(b, off); o.add_reads(b, off)
        assert g.counters()["n_minimizers"] == o.counters()["n_minimizers"], trial
        assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32)), trial
        g.close(); o.close()


@pytest.mark.parametrize("w", [3, 4, 7, 9, 12, 16])
@pytest.mark.parametrize("k", [21, 24, 28, 31])
def test_every_short_read_kernel_instance(k, w):
    """One case per instantiation of k_minimizer_fast (block size 4 / 9 / 16 x {w equal to it or not} x {k = 21, k <= 27,
    k >= 28, k = 31} x {one group per read, two}), with and without sketching intervals: minimizer count, k-mer spectrum
    and sketch against the oracle.  Regression for a code-generation accident found by tools/fuzz_parity.py (FUZZ_BIG_K):
    in the <16, k >= 28, w != 16, two groups> instance hipcc lost the wave index across the main loop, the block's
    minimizer count went to no LDS address at all and `n_minimizers` came back as stale LDS contents (spectrum and sketch
    were right, so tests that compared only those passed)."""
    rng = np.random.default_rng(100 * k + w)
    shapes = [(max(k + w - 1, min(256, k + 16 * w - 1) - 40), min(256, k + 16 * w - 1))]
    if 16 * w + k - 1 <= 256:
        shapes.append((k + 16 * w, min(512, k + 31 * w)))
    for lens in shapes:
        seqs = random_reads(rng, 700, lens)
        seqs[3] = seqs[3][:40] + b"N" + seqs[3][41:]                  # one deferred read
        for interval in (0, 250):
            o, g = run_both(seqs, k, w, 4, interval=interval, num_bins=4096, batches=2 if interval else 1)
            if not interval:
                assert np.array_equal(g.histogram(), o.histogram().astype(np.uint32)), (lens, interval)
            o.finish(); g.finish()
            oc, gc = o.counters(), g.counters()
            assert all(oc[key] == gc[key] for key in ("n_reads", "n_minimizers", "total_len")), (lens, interval, oc, gc)
            assert np.array_equal(g.cms(), o.cms()), (lens, interval)
            assert_same_sketch(o, g)
            g.close(); o.close()


def test_bin_reads_at_first_spectrum_any_order():
    """hulk_bin_reads_device_at: the spectra of a batch filled by separate calls, out of order (what a rank that bins whole
    intervals of a batch does for its own ones), then one flush of the batch — the sketch of the interval rule over the
    same reads, and refusals for spectra past the batch."""
    import torch
    h = gpu()
    k, w, S, I, L = 15, 9, 16, 700, 120
    rng = np.random.default_rng(5)
    seqs = random_reads(rng, 4 * I, L)
    bases, offsets = pack_reads(seqs)
    o = pyorc.Sketcher(k, w, S, 0, 1.0, I); o.add_reads(bases, offsets); o.finish()
    g = h.GpuSketcher(k, w, S, interval=0)
    db = torch.from_numpy(np.concatenate([bases, np.zeros(16, np.uint8)])).cuda()
    for first_spec, n_spec in ((2, 2), (0, 1), (1, 1)):
        lo = first_spec * I
        off = torch.from_numpy((offsets[lo:lo + n_spec * I + 1]).astype(np.int64)).cuda()
        g.bin_reads_device(db.data_ptr(), off.data_ptr(), n_spec * I, L, db.numel(), reads_per_spectrum=I, first_spectrum=first_spec)
    with pytest.raises(h.HulkError):
        g.bin_reads_device(db.data_ptr(), off.data_ptr(), I, L, db.numel(), reads_per_spectrum=I, first_spectrum=g.batch_size)
    with pytest.raises(h.HulkError):
        g.flush_batch(3)                                   # four spectra were filled
    g.flush_batch(4)
    g.finish()
    assert g.counters()["n_minimizers"] == o.counters()["n_minimizers"]
    assert np.array_equal(g.cms(), o.cms())
    assert_same_sketch(o, g)
    g.close(); o.close()


def test_two_vector_scan_gives_the_tile_minima_of_the_per_interval_products():
    """k_cws_scan<2> forms min_t K * rcp_t as min(K * max_t rcp_t, K * min_t rcp_t) (a rounded fp32 product is monotone in either
    factor): its tile minima must equal, bit for bit, those of the loop over the batch's T reciprocal vectors (k_cws_scan<1>,
    HULK_SCAN_MERGE_LOOP in the profiling build) — and the sketch that of the per-interval path (HULK_SCAN_PER_INTERVAL).
    The switches are read once per process: one child process per variant, every tile read (HULK_FLAG_NO_PRUNE)."""
    import subprocess
    import sys
    import tempfile
    exp = os.path.join(ROOT, "hulk_amd", "csrc", "libhulkhip_exp.so")
    if not os.path.exists(exp):
        pytest.skip("profiling build (make -C hulk_amd/csrc EXPERIMENTS=1) not present")
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, hulk_amd\n"
        "from hulk_amd import _lib, synth\n"
        "bases, offsets = synth.reads_numpy(3, 27500, 150)\n"
        "g = hulk_amd.GpuSketcher(17, 9, 40, interval=2000, batch=8, flags=_lib.HULK_FLAG_NO_PRUNE)\n"
        "cut = 16000                                  # exactly one batch of 8 intervals: flushed by this call\n"
        "g.add_reads(bases[:cut * 150], offsets[:cut + 1])\n"
        "tm, sm = g.debug_read(_lib.HULK_DEBUG_TILEMIN), g.debug_read(_lib.HULK_DEBUG_SCANMAP)\n"
        "g.add_reads(bases[cut * 150:], offsets[cut:] - offsets[cut]); g.finish()\n"
        "m, w = g.sketch()\n"
        "np.savez(sys.argv[1], mins=m, weights=w, tilemin=tm, scanmap=sm)\n"
        "g.close()\n")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for label, extra in (("two_vector", {}), ("loop", {"HULK_SCAN_MERGE_LOOP": "1"}), ("per_interval", {"HULK_SCAN_PER_INTERVAL": "1"}),
                             ("grid", {"HULK_SCAN_GRID": "1"})):
            p = os.path.join(td, label + ".npz")
            r = subprocess.run([sys.executable, "-c", code, p], env=dict(os.environ, HULK_LIB="exp", **extra), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            out[label] = dict(np.load(p))
    a, b = out["two_vector"], out["loop"]
    assert np.array_equal(a["scanmap"], b["scanmap"]) and a["scanmap"].any()
    assert a["tilemin"].view(np.uint32).tolist() == b["tilemin"].view(np.uint32).tolist() or np.array_equal(a["tilemin"], b["tilemin"], equal_nan=True)
    assert np.isfinite(a["tilemin"]).any()
    assert np.array_equal(out["grid"]["tilemin"], a["tilemin"], equal_nan=True)       # list form == grid form
    for label in ("loop", "per_interval", "grid"):
        assert np.array_equal(out[label]["mins"], a["mins"]) and np.array_equal(out[label]["weights"], a["weights"]), label


def test_profile_table_times_every_launch_of_a_one_stream_context():
    """hulk_set_profiling bit 32 + hulk_get_profile_table (ABI 4): on a HULK_FLAG_NO_OVERLAP context every launch of the binning chain
    and of the flush is timed once per batch, the durations are positive and add up to less than the wall clock, and the switch changes
    no result."""
    import time
    from hulk_amd import _lib, synth
    bases, offsets = synth.reads_numpy(0, 24_000, 150)
    out = {}
    for prof in (0, 32):
        g = gpu().GpuSketcher(15, 9, 16, interval=2000, batch=4, flags=_lib.HULK_FLAG_NO_OVERLAP)
        g.set_profiling(prof)
        t0 = time.perf_counter()
        g.add_reads(bases, offsets)                              # 12 intervals = 3 batches of 4
        g.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        tbl = g.profile_table() if prof else {}
        g.finish()
        out[prof] = (g.sketch(), g.cms(), tbl, wall_ms)
        g.close()
    assert np.array_equal(out[0][0][0], out[32][0][0]) and np.array_equal(out[0][0][1], out[32][0][1]) and np.array_equal(out[0][1], out[32][1])
    tbl, wall_ms = out[32][2], out[32][3]
    for kname in ("k_minimizer_fast", "k_region_offsets", "k_jump_bin", "k_nibble_hist", "k_nibble_merge", "k_count_used", "k_flush_decide",
                  "k_cms_segsum", "k_cms_base", "k_cms_freq", "k_scan_test", "k_cws_apply"):
        assert kname in tbl and tbl[kname][0] >= 3 and tbl[kname][1] > 0, (kname, tbl.get(kname))
    assert tbl["k_cms_freq"][0] == tbl["k_flush_decide"][0] == 3                      # one flush per batch of 4 intervals
    assert "-" not in tbl and 0 < sum(v[1] for v in tbl.values()) < wall_ms
    assert out[0][2] == {}


def test_long_sequences_on_two_work_lanes_share_one_scratch_in_order():
    """The long-sequence path returns with its kernels queued (round 6; it used to wait for them), and consecutive batches run on
    alternating work lanes — while the descriptors and the per-sequence set tables on the device are one set per context.  A group
    queued on one lane must wait for the group queued before it on the other (tools/fuzz_parity.py found the race: seed 7, case 3).
    Many intervals of one batch each, every read long: sketch and minimizer count against the oracle.  (With the ordering switched
    off — HULK_LONG_NO_LANE_ORDER, profiling build — this test fails.)"""
    rng = np.random.default_rng(99)
    k, w, S, I = 15, 5, 32, 16
    seqs = random_reads(rng, 400, (20_000, 40_000), b"ACGTN")      # a group of 16 such reads keeps the chip busy for ~100 us
    o = pyorc.Sketcher(k, w, S, 0, 0.02, I)
    g = gpu().GpuSketcher(k, w, S, I, 0.02, 0, batch=1, work_lanes=2)
    bases, offsets = pack_reads(seqs)
    o.add_reads(bases, offsets)
    for a in range(0, len(seqs), 80):                              # several calls: batches keep alternating lanes across them
        g.add_reads(bases, offsets[a:min(a + 80, len(seqs)) + 1])
    o.finish(); g.finish()
    assert_same_sketch(o, g)
    assert o.counters()["n_minimizers"] == g.counters()["n_minimizers"]
    g.close(); o.close()
