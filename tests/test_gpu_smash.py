"""`hulk smash` on the GPU vs the literal CPU restatement: distances must be
bit-identical (the kernel accumulates over the slots in the reference's order)."""
import numpy as np
import pytest

from oracle import pyorc

pytestmark = pytest.mark.gpu


def make_sketches(rng, n, s, bins=194481, related=True):
    base = rng.integers(0, bins, size=s).astype(np.uint64)
    mins = np.empty((n, s), dtype=np.uint64)
    for i in range(n):
        keep = rng.random(s) < (0.2 + 0.6 * rng.random()) if related else np.zeros(s, bool)
        mins[i] = np.where(keep, base, rng.integers(0, bins, size=s).astype(np.uint64))
    weights = -rng.gamma(2.0, 1e-3, size=(n, s))            # histosketch weights are mostly negative
    weights[rng.random((n, s)) < 0.05] *= -1                # ... some positive
    return mins, weights


@pytest.mark.parametrize("n,s", [(2, 50), (17, 512), (65, 2048), (33, 100)])
def test_distance_matrix_bit_exact(n, s):
    from hulk_amd.smash import distance_matrix
    rng = np.random.default_rng(n * 7 + s)
    mins, weights = make_sketches(rng, n, s)
    for metric in ("jaccard", "weightedjaccard"):
        got = distance_matrix(mins, weights, metric)
        want = pyorc.smash_matrix(mins, weights, metric)
        assert np.array_equal(got, want), metric
    # the subject-weights quirk (sketchio.go:296) makes the weighted matrix asymmetric; jaccard is symmetric
    j = distance_matrix(mins, weights, "jaccard")
    assert np.array_equal(j, j.T) and np.all(np.diag(j) == 0)


def test_untouched_slots_and_special_values():
    """MaxFloat64 weights (slots never updated) overflow the union: Inf/Inf = NaN, as in Go."""
    from hulk_amd.smash import distance_matrix, go_format_f2
    mins = np.array([[1, 2, 3, 4], [1, 2, 9, 4], [0, 0, 0, 0]], dtype=np.uint64)
    w = np.array([[-1.0, 2.0, -3.0, 4.0], [np.finfo(np.float64).max] * 4, [0.0, -0.0, 0.0, 0.0]])
    got = distance_matrix(mins, w, "weightedjaccard")
    want = pyorc.smash_matrix(mins, w, "weightedjaccard")
    assert np.array_equal(got, want, equal_nan=True)
    assert np.isnan(got[1, 0]) and np.isnan(got[2, 2])
    assert go_format_f2(float("nan")) == "NaN" and go_format_f2(100 - 0.125 * 100) == "87.50"


def test_smash_cli_end_to_end(tmp_path):
    """sketch three read sets with the GPU path, smash them, compare the CSV with the oracle's matrix."""
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd.__main__ import main
    from hulk_amd.sketchio import HULKdata
    from hulk_amd.smash import go_format_f2
    d = tmp_path / "sk"
    d.mkdir()
    mins, weights, names = [], [], []
    for i, (first, n) in enumerate(((0, 3000), (1000, 3000), (50000, 2500))):
        g = hulk_amd.GpuSketcher(15, 9, 64, interval=1000)
        g.add_reads(*synth.reads_numpy(first, n, 120))
        g.finish()
        hs = g.histosketch()
        doc = HULKdata(); doc.add(hs); doc.filename = f"r{i}.fq,"; doc.banner_label = "blank"
        p = d / f"s{2 - i}.json"          # file order != creation order: the header must be sorted
        doc.write_json(p)
        names.append(str(p)); mins.append(hs.mins); weights.append(hs.weights)
        g.close()
    out = str(tmp_path / "res")
    for metric in ("jaccard", "weightedjaccard"):
        assert main(["smash", "-d", str(d), "-k", "15", "-m", metric, "-o", out]) == 0
        rows = open(out + ".hulk-matrix.csv").read().splitlines()
        order = sorted(names)
        assert rows[0] == ",".join(order)
        idx = [names.index(f) for f in order]
        want = pyorc.smash_matrix(np.stack(mins)[idx], np.stack(weights)[idx], metric)
        for r, line in enumerate(rows[1:]):
            assert line == ",".join(go_format_f2(100 - v * 100) for v in want[r])
    # --bannerMatrix (cmd/smash.go:229-261): mins of every sketch + its banner label
    assert main(["smash", "-d", str(d), "-k", "15", "-o", out, "--bannerMatrix"]) == 0
    brow = open(out + ".banner-matrix.csv").read().splitlines()
    assert len(brow) == 3
    for line, f in zip(brow, sorted(names)):
        assert line == ",".join(str(int(v)) for v in mins[names.index(f)]) + ",blank"
    assert main(["smash", "-d", str(d), "-m", "euclidean", "-o", out]) == 1     # not in availMetrics
    assert main(["smash", "-d", str(d), "-k", "21", "-o", out]) == 1            # no sketch with that k
    # the native form (hulk_smash_files: loader, ordering and CSV in the library) writes the bytes the Python form wrote until
    # round 6 (json + hashlib + "%.2f"), matrix and banner file alike, and a file name that encoding/csv must quote
    from hulk_amd import smash as smash_mod
    odd = d / 'a,"b".json'
    odd.write_text((d / "s0.json").read_text())
    for metric in ("jaccard", "weightedjaccard"):
        o1, m1 = smash_mod.smash(str(d), out + ".native", 15, "histosketch", metric, banner_matrix=True)
        o2, m2 = smash_mod.smash_python(str(d), out + ".python", 15, "histosketch", metric, banner_matrix=True)
        assert o1 == o2 and len(o1) == 4 and np.array_equal(m1, m2, equal_nan=True)
        for suffix in (".hulk-matrix.csv", ".banner-matrix.csv"):
            assert open(out + ".native" + suffix, "rb").read() == open(out + ".python" + suffix, "rb").read()
    assert open(out + ".native.hulk-matrix.csv").readline().startswith('"' + str(d) + '/a,""b"".json",')
    # LoadHULKdata's checks at the command's level (sketchio.go:171-193, 243-254): corrupted MD5, another version, duplicate k
    import json
    from hulk_amd._lib import HulkError
    raw = json.loads((d / "s0.json").read_text())
    for name, edit, text in (("zz_md5.json", lambda x: x["signatures"][0]["Sketch"]["mins"].__setitem__(0, x["signatures"][0]["Sketch"]["mins"][0] ^ 1), "md5sum mismatch: "),
                             ("zz_ver.json", lambda x: x.__setitem__("version", "0.0.1"), "the loaded sketch was created with a different version of HULK: 0.0.1"),
                             ("zz_dup.json", lambda x: x["signatures"].append(x["signatures"][0]), "found 2 possible duplicate sketches in the supplied sketch file: ")):
        x = json.loads(json.dumps(raw)); edit(x)
        (d / name).write_text(json.dumps(x, indent=4))
        with pytest.raises(HulkError, match=text):
            smash_mod.smash(str(d), out + ".bad", 15)
        assert main(["smash", "-d", str(d), "-k", "15", "-o", out + ".bad"]) == 1
        (d / name).unlink()


def test_c5_full_size_sampled_against_oracle():
    """BASELINE config C5 at its stated size: 1024 sketches x sketchSize 2048, both metrics (cmd/smash.go:183-226,
    sketchio.go:259-306).  A pair's distance depends on its two sketches only, so the oracle's matrix of 64 random
    sketches must equal the corresponding 64 x 64 sub-matrix of the GPU's 1024 x 1024 one, bit for bit."""
    from hulk_amd.smash import distance_matrix
    rng = np.random.default_rng(1024)
    N, S = 1024, 2048
    mins, weights = make_sketches(rng, N, S)
    idx = np.sort(rng.choice(N, size=64, replace=False))
    for metric in ("weightedjaccard", "jaccard"):
        full = distance_matrix(mins, weights, metric)
        assert full.shape == (N, N)
        want = pyorc.smash_matrix(mins[idx], weights[idx], metric)
        assert np.array_equal(full[np.ix_(idx, idx)], want), metric
        if metric == "jaccard":
            assert np.array_equal(full, full.T) and np.all(np.diag(full) == 0)
