"""BASELINE.json's full-size configurations on the GPU, checked through size-independent properties
(the oracle needs hours at these sizes): invariance of the sketch under the interval-batch size and
under the way reads are cut into calls, linearity of the k-mer spectrum, counters, determinism, and a
prefix of the stream against the oracle.

C2: 10^7 reads x 150 bp, k=21, w=9, sketchSize=512, interval=100k.
C3-shaped: k=31, sketchSize=1024, decay on (22.7 GB of tables) on 5*10^6 reads.
"""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L = 150


def gpu():
    import hulk_amd
    return hulk_amd


def _run_stream(k, S, interval, decay, n_reads, chunk, batch, lanes=0):
    """Sketch synthetic reads [0, n_reads) fed as device-resident chunks of `chunk` reads."""
    import torch
    from hulk_amd import synth
    g = gpu().GpuSketcher(k, 9, S, interval, decay, batch=batch, work_lanes=lanes)
    assert g.batch_size == batch
    first = 0
    while first < n_reads:
        n = min(chunk, n_reads - first)
        b, off = synth.reads_torch(first, n, L)
        torch.cuda.synchronize()            # generated on torch's stream; the context runs on its own
        g.add_reads_device(b.data_ptr(), off.data_ptr(), n, L, b.numel())
        torch.cuda.synchronize()            # b/off are released when they go out of scope
        first += n
    g.finish()
    m, w = g.sketch()
    c = g.counters()
    g.close()
    return m, w, c


def test_c2_full_size_invariances():
    n = 10_000_000
    m1, w1, c1 = _run_stream(21, 512, 100_000, 1.0, n, 1_600_000, 16)
    assert c1["n_reads"] == n and c1["total_len"] == n * L
    assert 26.5 * n < c1["n_minimizers"] < 27.5 * n           # ~27.0 distinct minimizers per random 150 bp read
    assert m1.max() < 21 ** 4
    assert np.all(np.isfinite(w1)) and np.all(w1 < np.finfo(np.float64).max)
    # other batch size, other call boundaries (not multiples of the interval): bit-identical sketch
    m2, w2, c2 = _run_stream(21, 512, 100_000, 1.0, n, 1_234_567, 5)
    assert np.array_equal(m1, m2) and np.array_equal(w1, w2) and c1 == c2
    # determinism of the pipeline (batches binned on two alternating work streams, flushed on a third)
    m3, w3, _ = _run_stream(21, 512, 100_000, 1.0, n, 1_600_000, 16)
    assert np.array_equal(m1, m3) and np.array_equal(w1, w3)
    # ... and one work lane computes the same
    m4, w4, c4 = _run_stream(21, 512, 100_000, 1.0, n, 1_600_000, 16, lanes=1)
    assert np.array_equal(m1, m4) and np.array_equal(w1, w4) and c1 == c4


def test_c2_prefix_against_oracle():
    """The first 8 intervals of the C2 stream (800k reads) against the CPU oracle, C2 parameters: the first batch
    evaluates everything, later intervals run through the per-tile and whole-batch bounds."""
    from oracle import pyorc
    from hulk_amd import synth
    n = 800_000
    m, w, c = _run_stream(21, 512, 100_000, 1.0, n, 170_000, 3)
    bases, offsets = synth.reads_numpy(0, n, L)
    o = pyorc.Sketcher(21, 9, 512, 0, 1.0, 100_000)
    o.add_reads(bases, offsets)
    o.finish()
    mo, wo = o.sketch()
    assert o.counters()["n_minimizers"] == c["n_minimizers"]
    assert np.array_equal(m, mo)
    assert np.allclose(w, wo, rtol=1e-9, atol=0)
    o.close()


def test_spectrum_linearity_full_interval():
    """hist(A u B) = hist(A) + hist(B) on two full 100k-read intervals, and sum(hist) = #minimizers."""
    import torch
    from hulk_amd import synth
    g = gpu().GpuSketcher(21, 9, 8)                    # no interval: everything stays in one spectrum
    parts = []
    for first, n in ((0, 100_000), (100_000, 100_000)):
        h = gpu().GpuSketcher(21, 9, 8)
        b, off = synth.reads_torch(first, n, L)
        torch.cuda.synchronize()
        h.add_reads_device(b.data_ptr(), off.data_ptr(), n, L, b.numel())
        g.add_reads_device(b.data_ptr(), off.data_ptr(), n, L, b.numel())
        torch.cuda.synchronize()
        parts.append(h.histogram().astype(np.uint64))
        assert int(parts[-1].sum()) == h.counters()["n_minimizers"]
        h.close()
    both = g.histogram().astype(np.uint64)
    assert np.array_equal(both, parts[0] + parts[1])
    assert int(both.sum()) == g.counters()["n_minimizers"]
    assert (both > 0).mean() > 0.99                    # ~all 194,481 bins used per interval
    g.close()


def test_c3_shape_drift_batch_invariance():
    """k=31, sketchSize=1024, concept drift on (decay 0.02): 22.7 GB of CWS tables resident; the sketch
    must not depend on how many intervals are flushed per pass over the table."""
    n = 5_000_000
    m1, w1, c1 = _run_stream(31, 1024, 100_000, 0.02, n, 1_600_000, 16)
    m2, w2, c2 = _run_stream(31, 1024, 100_000, 0.02, n, 900_001, 3, lanes=1)
    assert c1 == c2 and c1["n_reads"] == n
    assert np.array_equal(m1, m2) and np.array_equal(w1, w2)
    assert m1.max() < 31 ** 4 and np.all(np.isfinite(w1))


def test_c3_full_size_50m_reads():
    """BASELINE config C3 as stated: 50 M reads x 150 bp, k = 31, sketchSize = 1024, concept drift on (decay 0.02),
    interval 100k = 500 sketching intervals over 22.7 GB of CWS tables.  The oracle would need days, so the run is
    checked through size-independent properties: counters, value ranges, and bit-identical sketches under a different
    interval-batch size and different call boundaries (the batch changes which intervals share a pass over the table
    and a count-min replay; the decay path is the order-dependent one)."""
    n = 50_000_000
    m1, w1, c1 = _run_stream(31, 1024, 100_000, 0.02, n, 1_600_000, 16)
    assert c1["n_reads"] == n and c1["total_len"] == n * L
    assert 24.6 * n < c1["n_minimizers"] < 25.6 * n           # ~25.1 distinct minimizers per read at k = 31
    assert m1.max() < 31 ** 4 and np.all(np.isfinite(w1))
    assert (w1 < 0).all()                                     # every slot has been taken by a negative weight long since
    assert len(np.unique(m1)) > 900                           # slots are independent samples of the spectrum
    m2, w2, c2 = _run_stream(31, 1024, 100_000, 0.02, n, 2_345_678, 7)
    assert c1 == c2
    assert np.array_equal(m1, m2) and np.array_equal(w1, w2)


@pytest.mark.parametrize("decay", [0.02, 0.5])
def test_drift_pruning_against_oracle_many_intervals(decay):
    """Concept drift with the scan pruned against w_start/decayWeight (negative weights only): 10 intervals of
    20k reads, k=17, against the oracle's element-by-element AddElement."""
    from oracle import pyorc
    from hulk_amd import synth
    n, interval = 200_000, 20_000
    m, w, c = _run_stream(17, 48, interval, decay, n, 70_001, 4)
    bases, offsets = synth.reads_numpy(0, n, L)
    o = pyorc.Sketcher(17, 9, 48, 0, decay, interval)
    o.add_reads(bases, offsets)
    o.finish()
    mo, wo = o.sketch()
    assert np.array_equal(m, mo)
    assert np.allclose(w, wo, rtol=1e-7, atol=0)
    assert (wo < 0).all()                      # the regime the drift pruning relies on
    o.close()


@pytest.mark.parametrize("extra", [1, 2, 3])
def test_host_chunk_of_8_mib_and_a_few_bytes_is_copied_whole(extra):
    """hulk_add_reads copies a chunk of >= 8 MB into pinned staging with four threads.  With pieces of floor(n / 4) bytes
    rounded up to 64 (as first written) the last n mod 4 bytes never reached the staging buffer whenever floor(n / 4) was a
    multiple of 64 — n = 8 MiB + 1..3 — and the chunk's last read was sketched with whatever the buffer held before
    (ADVICE r3).  Both staging sets are dirtied with other reads first; the device-resident path is the comparison."""
    import torch
    nbytes = (8 << 20) + extra
    assert (nbytes // 4) % 64 == 0 and nbytes % 4 == extra
    rng = np.random.default_rng(80 + extra)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    n = nbytes // 150
    lens = np.full(n, 150, dtype=np.uint64)
    lens[-1] += nbytes - int(lens.sum())                          # the last read takes the remainder: its last bases are the bytes at stake
    offsets = np.zeros(n + 1, dtype=np.uint64); offsets[1:] = np.cumsum(lens)
    real = acgt[rng.integers(0, 4, nbytes)]
    filler = acgt[rng.integers(0, 4, nbytes)]
    filler[-4:] = np.where(real[-4:] == ord("A"), ord("C"), ord("A")).astype(np.uint8)   # what a stale tail would look like
    g = gpu().GpuSketcher(21, 9, 4)
    d = gpu().GpuSketcher(21, 9, 4)
    for data in (filler, filler, real):                           # two staging sets: both hold the filler when `real` arrives
        g.add_reads(data, offsets)
        tb, to = torch.from_numpy(data).cuda(), torch.from_numpy(offsets.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        d.add_reads_device(tb.data_ptr(), to.data_ptr(), n, int(lens.max()), nbytes)
        d.synchronize()
    assert np.array_equal(g.histogram(), d.histogram())
    assert g.counters() == d.counters()
    g.close(); d.close()


def test_host_buffers_in_chunks_equal_device_path():
    """hulk_add_reads with 1.3 M reads in ONE call: three chunks through the two pinned staging sets (chunk borders
    inside intervals) — same sketch and counters as the device-resident path."""
    from hulk_amd import synth
    n = 1_300_000
    m, w, c = _run_stream(21, 64, 100_000, 1.0, n, 1_300_000, 16)
    g = gpu().GpuSketcher(21, 9, 64, 100_000, 1.0, batch=16)
    bases, offsets = synth.reads_numpy(0, n, L)
    g.add_reads(bases, offsets)
    bases[:] = 0                                   # the call has copied the buffers: scribbling over them changes nothing
    g.finish()
    m2, w2 = g.sketch()
    assert g.counters() == c
    assert np.array_equal(m, m2) and np.array_equal(w, w2)
    g.close()


def test_long_sequences_against_oracle_and_split_invariance():
    """bench.py's `long_reads` shapes (5 kb reads, 500 kb FASTA contigs: src/pipeline/sketch.go:102-135 hands the whole record to
    AddSeq) run k_minimizer_bin's deferral + k_long_tile.  A prefix against the oracle — spectrum, minimizer count,
    sketch — and, at the leg's full size, properties that do not need it: the k-mer spectrum is a sum over sequences (two halves
    binned separately add up to the whole), every sequence is counted once, the result does not depend on how calls cut the stream."""
    import torch
    from hulk_amd import synth
    from oracle import pyorc
    # (1) prefix vs the oracle: 300 reads of 5 kb, one contig of 500 kb (interval 0: one spectrum, one flush)
    for n, Lx in ((300, 5_000), (1, 500_000)):
        bases, offsets = synth.reads_numpy(0, n, Lx)
        g = gpu().GpuSketcher(21, 9, 64)
        g.add_reads(bases, offsets)
        gh = g.histogram()
        g.finish()
        o = pyorc.Sketcher(21, 9, 64, 0, 1.0, 0)
        o.add_reads(bases, offsets)
        oh = o.histogram().astype(np.uint32)
        o.finish()
        assert np.array_equal(gh, oh), f"spectra differ in {(gh != oh).sum()} bins"
        assert g.counters()["n_minimizers"] == o.counters()["n_minimizers"]
        gm, gw = g.sketch(); om, ow = o.sketch()
        assert np.array_equal(gm, om) and np.allclose(gw, ow, rtol=1e-9, atol=0)
        g.close(); o.close()
    # (2) the leg's sizes: linearity of the spectrum and call-boundary invariance
    for n, Lx in ((200_000, 5_000), (2_000, 500_000)):
        b, off = synth.reads_torch(0, n, Lx)
        torch.cuda.synchronize()
        hists, cnts = [], []
        for cuts in ((0, n), (0, n // 2, n), (0, n // 3 + 1, n)):
            g = gpu().GpuSketcher(21, 9, 8)
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                sub_off = (off[lo:hi + 1] - off[lo]).contiguous()
                sub = b[int(off[lo]):]
                torch.cuda.synchronize()
                g.add_reads_device(sub.data_ptr(), sub_off.data_ptr(), hi - lo, Lx, sub.numel())
                g.synchronize()
            hists.append(g.histogram()); cnts.append(g.counters())
            g.close()
        assert np.array_equal(hists[0], hists[1]) and np.array_equal(hists[0], hists[2])
        assert cnts[0] == cnts[1] == cnts[2] and cnts[0]["n_reads"] == n and cnts[0]["total_len"] == n * Lx
        assert int(hists[0].sum()) == cnts[0]["n_minimizers"]                    # every distinct minimizer of a sequence is one increment
        assert 0.15 * n * Lx < cnts[0]["n_minimizers"] < 0.25 * n * Lx           # ~2 / (w + 1) per position
