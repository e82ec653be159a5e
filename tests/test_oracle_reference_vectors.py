"""Pins the CPU oracle against every vector the reference's own tests hold for this path, the
published vectors of its un-vendored dependencies, and known answers of Go's math/rand.
Runs on CPU (no GPU needed)."""
import hashlib

import numpy as np
import pytest

from oracle import pyorc


def test_nt4_table_reference_test():
    # src/minimizer/minimizer_test.go:14-30: "ACGTN" -> 0,1,2,3,4
    for ch, want in zip(b"ACGTN", (0, 1, 2, 3, 4)):
        assert pyorc.nt4(ch) == want
    # the full 256-entry table (minimizer.go:13-30)
    want = [4] * 256
    want[0:4] = [0, 1, 2, 3]
    for chs, v in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"TtUu", 3)):
        for ch in chs:
            want[ch] = v
    assert [pyorc.nt4(i) for i in range(256)] == want


def test_minimizer_reference_test_input():
    # src/minimizer/minimizer_test.go:32-61 sketches "ACTGAAAATTTT" (k=4,w=4) twice and requires
    # equal sets; the expected set itself is cross-checked against SURVEY.md App. C
    a = set(pyorc.minimizers(b"ACTGAAAATTTT", 4, 4).tolist())
    b = set(pyorc.minimizers(b"ACTGAAAATTTT", 4, 4).tolist())
    assert a == b == {3076, 7425, 17156}


def test_minimizer_parameter_errors():
    # minimizer.go:62-76
    with pytest.raises(pyorc.OracleError, match="w must be"):
        pyorc.minimizers(b"A" * 400, 5, 257)
    with pytest.raises(pyorc.OracleError, match="k size must be"):
        pyorc.minimizers(b"A" * 400, 32, 5)
    with pytest.raises(pyorc.OracleError, match="sequence length must be > 0"):
        pyorc.minimizers(b"", 5, 5)
    with pytest.raises(pyorc.OracleError, match=">= w \\+ k - 1"):
        pyorc.minimizers(b"ACGTACGT", 5, 5)
    assert len(pyorc.minimizers(b"ACGTACGTA", 5, 5)) >= 1


def test_pow_reference_test():
    # src/helpers/helpers_test.go:7-17
    assert [pyorc.ipow(2, b) for b in (2, 3, 4)] == [4, 8, 16]
    assert pyorc.ipow(21, 4) == 194481 and pyorc.ipow(31, 4) == 923521


def test_kmerspectrum_reference_test():
    # src/kmerspectrum/kmerspectrum_test.go:14-45: negative bins rejected; AddHash(1), AddHash(1234)
    # into 10 bins -> cardinality 1 then 2
    with pytest.raises(pyorc.OracleError, match="negative value"):
        pyorc.Sketcher(4, 4, 2, num_bins=-1)
    assert pyorc.jump(1, 10) != pyorc.jump(1234, 10)
    s = pyorc.Sketcher(4, 4, 2, num_bins=10)
    assert s.B == 10 and s.used_bins() == 0
    h = np.zeros(10, dtype=np.uint32); h[pyorc.jump(1, 10)] += 1
    s.add_histogram(h); assert s.used_bins() == 1
    h[:] = 0; h[pyorc.jump(1234, 10)] += 1
    s.add_histogram(h); assert s.used_bins() == 2


def test_jump_hash_published_vectors():
    # github.com/dgryski/go-jump @ e1f439676b57 (jump_test.go) / Lamping & Veach reference code
    for key, n, want in ((1, 1, 0), (42, 57, 43), (0xDEAD10CC, 1, 0), (0xDEAD10CC, 666, 361),
                         (256, 1024, 520)):
        assert pyorc.jump(key, n) == want
    assert pyorc.jump(12345, 0) == 0 and pyorc.jump(12345, -5) == 0   # n <= 0 -> 1 bucket


def test_jump_hash_consistency_property():
    # defining property: growing n only moves keys into the new bucket
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 2 ** 63, size=2000).tolist()
    for n in (10, 999, 194481):
        a = [pyorc.jump(k, n) for k in keys]
        b = [pyorc.jump(k, n + 1) for k in keys]
        assert all(x == y or y == n for x, y in zip(a, b))
        assert all(0 <= x < n for x in a)


def test_hash64_cross_check_values():
    # SURVEY.md App. C (independent restatement) + invertibility (minimap2 hash64 is a bijection)
    m42 = (1 << 42) - 1
    assert [pyorc.hash64(x, m42) for x in (0, 1, 2, 0x2AAAAAAAAAA, 123456789)] == \
        [2057405897664, 454051559630, 3570982848116, 1728553308967, 900621528218]
    m62 = (1 << 62) - 1
    assert [pyorc.hash64(x, m62) for x in (0, 1, 123456789)] == \
        [2158324264573792932, 2002549777813010638, 2746896918742293240]
    m8 = (1 << 8) - 1
    assert sorted(pyorc.hash64(x, m8) for x in range(256)) == list(range(256))


def test_go_math_rand_known_answers():
    # Go (<1.20) math/rand with Seed(1): the stream every un-seeded Go program printed
    g = pyorc.GoRand(1)
    assert [g.int63() for _ in range(10)] == [
        5577006791947779410, 8674665223082153551, 6129484611666145821, 4037200794235010051,
        3916589616287113937, 6334824724549167320, 605394647632969758, 1443635317331776148,
        894385949183117216, 2775422040480279449]
    g = pyorc.GoRand(1)
    assert [g.float64() for _ in range(5)] == [
        0.6046602879796196, 0.9405090880450124, 0.6645600532184904, 0.4377141871869802,
        0.4246374970712657]
    # rand.Intn(100) after Seed(1): 81 87 47 59 81 18 25 40 56 0  (Int31n(100) = Int63()>>32 % 100 path)
    g = pyorc.GoRand(1)
    got = []
    for _ in range(10):
        v = g.int63() >> 32                       # Int31()
        # Int31n: n=100 is not a power of two -> rejection on max = (1<<31)-1 - (1<<31)%100
        mx = (1 << 31) - 1 - (1 << 31) % 100
        while v > mx:
            v = g.int63() >> 32
        got.append(v % 100)
    assert got == [81, 87, 47, 59, 81, 18, 25, 40, 56, 0]


def test_gamma_uniform_moments():
    # go_rng restatement is unpinned against Go output; at least the distributions are right
    S, B = 4, 50000
    r, c, b = pyorc.cws_tables(S, B)
    assert abs(r.mean() - 2.0) < 0.02 and abs(r.var() - 2.0) < 0.06          # Gamma(2,1)
    assert abs(np.exp(c).mean() - 2.0) < 0.02
    beta = b / r
    assert abs(beta.mean() - 0.5) < 0.005 and beta.min() >= 0 and beta.max() < 1
    # the uniform generator shares seed 1 with the gamma generator: first beta = first Float64
    assert beta[0, 0] == pytest.approx(0.6046602879796196, rel=1e-15)


def test_cms_geometry_and_positions():
    # countmin.go:31-32 -> 7 x 2000; SURVEY.md App. C positions for bin 5
    assert pyorc.cms_geometry() == (7, 2000)
    assert [pyorc.jump((5 * (d + 1)) & (2 ** 64 - 1), 2000) for d in range(7)] == \
        [1881, 751, 21, 1342, 1152, 1241, 1087]


def test_fixture_spectrum(fq_reads):
    """The reference's CI smoke input (.travis.yml:22): must use >= 1 % of the bins; spectrum
    statistics agree with the independent probe of SURVEY.md App. C."""
    assert len(fq_reads) == 1000 and all(len(r) == 100 for r in fq_reads)
    first = pyorc.minimizers(fq_reads[0], 21, 9)
    assert len(first) == 15 and sorted(first.tolist())[:3] == [24527002630165, 56854662392589, 73925710344725]
    for k, tot, used, mx, sha in (
            (21, 17040, 12212, 15, "d4e4bf949482bdf426dd821c4bbe924ba9e395de5e4e9accc736bce888049de1"),
            (31, 15033, 11702, 14, "732d2e405086edec32fb31204ee01ad8a83856669461f7b2ae05a11866ff6bf0")):
        s = pyorc.Sketcher(k, 9, 2)
        for r in fq_reads:
            s.add_read(r)
        h = s.histogram()
        assert s.counters()["n_minimizers"] == tot and s.used_bins() == used and h.max() == mx
        assert hashlib.sha256(h.astype("<u4").tobytes()).hexdigest() == sha
        assert used / k ** 4 >= 0.01
        s.finish()


def test_few_bins_is_fatal_and_empty_is_skipped():
    s = pyorc.Sketcher(21, 9, 2)
    s.add_read(b"ACGTTGCATGCATGCAAAGTCGATCGATCGGGCTAGCTAGCTAGCTTTGAC")
    with pytest.raises(pyorc.OracleError, match="not used yet"):   # kmerspectrum.go:94-96
        s.flush()
    s2 = pyorc.Sketcher(21, 9, 2)
    s2.flush()                                                      # boss.go:118: nothing to do
    with pytest.raises(pyorc.OracleError, match="no sequences received"):
        s2.finish()


def test_addelement_literal_small_case():
    """AddElement (histosketch.go:129-155) recomputed in numpy from the same CWS tables."""
    k, S = 5, 6
    B = k ** 4
    s = pyorc.Sketcher(k, 3, S)
    r, c, b = s.cws()
    rng = np.random.default_rng(5)
    hist = (rng.random(B) < 0.3) * rng.integers(1, 9, size=B)
    s.add_histogram(hist.astype(np.uint32)); s.flush()
    mins, weights = s.sketch()
    ctr = np.zeros((7, 2000)); wm = np.full(S, np.finfo(np.float64).max); mm = np.zeros(S, np.uint64)
    for bin_ in np.nonzero(hist)[0]:
        est = np.inf
        for d in range(7):
            g = pyorc.jump((int(bin_) * (d + 1)) & (2 ** 64 - 1), 2000)
            ctr[d, g] += hist[bin_]; est = min(est, ctr[d, g])
        A = c[:, bin_] / (np.exp(np.log(est) - b[:, bin_]) * np.exp(r[:, bin_]))
        upd = A < wm
        wm[upd] = A[upd]; mm[upd] = bin_
    assert np.array_equal(mm, mins) and np.allclose(wm, weights, rtol=1e-12)
    assert np.array_equal(ctr, s.cms())


def test_golden_sketch_reproduced_by_oracle(fq_reads):
    """tests/golden/c1_fixture_k15_s64_drift.json (tools/make_golden.py): the oracle still produces the
    committed sketch — guards the checker itself against silent changes (k=15 keeps the tables small)."""
    import os
    from conftest import GOLDEN
    from hulk_amd.sketchio import load_hulk_data
    g = load_hulk_data(os.path.join(GOLDEN, "c1_fixture_k15_s64_drift.json")).signatures[0][1]
    o = pyorc.Sketcher(15, 9, 64, 0, 0.05, 250)
    for r in fq_reads:
        o.add_read(r)
    o.finish()
    m, w = o.sketch()
    assert np.array_equal(m, g.mins) and np.array_equal(w, g.weights)     # JSON floats round-trip exactly
    assert g.concept_drift is True and g.num_histogram_bins == 50625


def test_gamma_squeeze_constant_is_immaterial():
    """SURVEY.md App. B flags go_rng's squeeze constant as its one uncertain recollection (4*exp(-0.5)/sqrt(2) vs
    CPython's 1 + ln 4.5).  The squeeze `r + M - 4.5 z >= 0` is only a shortcut that implies the real test
    `r >= ln z` for any M <= 1 + ln 4.5, so the accepted variates — hence r, c, b of newCWS — do not depend on it:
    both constants, and no squeeze at all, give the identical stream over 10^6 draws."""
    def draws(variant, n):
        pyorc.set_gamma_variant(variant)
        try:
            g = pyorc.GoRand(1)
            return np.array([g.gamma(2.0, 1.0) for _ in range(n)])
        finally:
            pyorc.set_gamma_variant(0)
    n = 1_000_000
    a = draws(0, n)
    assert abs(a.mean() - 2.0) < 0.01 and abs(a.var() - 2.0) < 0.03      # Gamma(2, 1)
    assert np.array_equal(a, draws(1, n))
    assert np.array_equal(a, draws(2, n))


def test_gamma_matches_cpython_gammavariate_on_the_go_stream():
    """go_rng's Gamma is (recalled) a port of CPython's random.gammavariate — Cheng's algorithm GB.
    CPython's own implementation is available here, so the restatement in oracle/hulk_oracle.c is pinned against it: the
    stdlib routine, fed the SAME uniform stream (Go math/rand Float64, seed 1) through a Random subclass, must return the
    oracle's variates bit for bit, rejections included (the stream position after N draws is compared too)."""
    import random

    class GoStream(random.Random):
        def __init__(self):
            super().__init__(0)
            self.src = pyorc.GoRand(1)
            self.calls = 0

        def random(self):
            self.calls += 1
            return self.src.float64()

    ref = GoStream()
    want = [ref.gammavariate(2.0, 1.0) for _ in range(200_000)]
    g = pyorc.GoRand(1)
    got = [g.gamma(2.0, 1.0) for _ in range(200_000)]
    assert got == want
    assert g.float64() == ref.src.float64()              # both consumed the same number of uniforms
    assert ref.calls > 2 * 200_000                        # ... and some attempts were rejected on the way
    # the first CWS parameters of HistoSketch.newCWS (histosketch.go:107-118) from that routine: r, c = ln(gamma), b = U * r
    r, c, b = pyorc.cws_tables(2, 16)
    ref2, uni = GoStream(), pyorc.GoRand(1)
    import math
    for i in range(2):
        for j in range(16):
            rr = ref2.gammavariate(2.0, 1.0); cc = math.log(ref2.gammavariate(2.0, 1.0)); bb = (0.0 + uni.float64() * 1.0) * rr
            assert r[i, j] == rr and c[i, j] == cc and b[i, j] == bb
