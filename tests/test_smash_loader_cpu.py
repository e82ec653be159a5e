"""hulk_load_sketches (hulk_amd/csrc/hulk_smashio.hip) — the native LoadHULKdata + FindSketch of `hulk smash`'s directory form
(src/sketchio/sketchio.go:100-257, cmd/smash.go:160-180) — against the Python loader (json + hashlib, hulk_amd/sketchio.py) on
the same files: values bit for bit, the reference's error texts, the order in which failures surface.  No GPU needed."""
import hashlib
import json
import os

import numpy as np
import pytest

from hulk_amd import smash as smash_mod
from hulk_amd._lib import HulkError
from hulk_amd.sketchio import HULKdata, HistoSketch, KHFSketch, load_hulk_data


def _write(path, mins, weights, k=21, banner="blank", filename="reads.fq,"):
    hs = HistoSketch(k, np.asarray(mins, dtype=np.uint64), np.asarray(weights, dtype=np.float64), k ** 4, False)
    d = HULKdata()
    d.filename, d.banner_label = filename, banner
    d.add(hs)
    d.write_json(path)
    return d


def _rand_sketch(rng, S):
    mins = rng.integers(0, 21 ** 4, size=S, dtype=np.uint64)
    w = rng.standard_normal(S) * 10.0 ** rng.integers(-12, 12, size=S)
    w[rng.integers(0, S)] = 1.7976931348623157e308          # an untouched slot (histosketch.go:84-87)
    w[rng.integers(0, S)] = 5e-324
    w[rng.integers(0, S)] = -0.0
    return mins, w


def test_native_loader_equals_python_loader(tmp_path):
    rng = np.random.default_rng(7)
    S = 257
    files = []
    for i in range(23):
        p = str(tmp_path / f"s{i:03d}.json")
        mins, w = _rand_sketch(rng, S)
        _write(p, mins, w, banner=f'lab,"{i}"' if i % 3 == 0 else f"b{i}")
        files.append(p)
    order, mins, weights, banners = smash_mod.load_sketches(list(reversed(files)) + files[:2], ksize=21, threads=5)   # any order, duplicates
    assert order == sorted(files)
    for i, p in enumerate(order):
        d = load_hulk_data(p)
        hs = d.signatures[0][1]
        assert np.array_equal(mins[i], hs.mins)
        ref = np.asarray(hs.weights, dtype=np.float64)
        # bit for bit, the denormal included — but for "-0": strconv.ParseFloat keeps the sign (as the library does), Python's json
        # reads the token as the integer 0
        zero = ref == 0
        assert weights[i][~zero].tobytes() == ref[~zero].tobytes() and not weights[i][zero].any()
        assert np.signbit(weights[i][zero]).sum() == 1
        assert banners[i] == d.banner_label


def test_native_loader_reads_what_encoding_json_reads(tmp_path):
    """Not only the writer's own layout: any spacing, key order, exponents, escapes, extra keys, duplicate keys (the last wins),
    case-insensitive field names of the Sketch object (encoding/json struct matching), several signatures."""
    mins = [3, 194480, 0, 77]
    md5 = hashlib.md5(np.asarray(mins, dtype="<u8").tobytes()).hexdigest()
    doc = {"banner_label": "café \U0001F600 \"q\"", "extra": {"a": [1, {"b": None}], "t": True}, "class": "hulk_sketch", "filename": "x",
           "hash_function": "ntHash", "license": "CC0", "version": "1.0.0",
           "signatures": [
               {"Algorithm": "khf", "Sketch": {"ksize": 21, "md5sum": md5, "mins": mins, "num": 4}},
               {"Sketch": {"KSize": 2.1e1, "MD5SUM": md5, "weights": [1e-3, -2.5E+2, 0.1, 17], "mins": [9], "Mins": [3.0, 1.9448e5, 0, 77],
                           "num": 4, "num_histogram_bins": 194481, "concept_drift": False}, "Algorithm": "histosketch"},
               {"Algorithm": "histosketch", "Sketch": {"ksize": 31, "md5sum": md5, "mins": mins, "weights": [1, 2, 3, 4]}}]}
    a, b = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    open(a, "w").write(json.dumps(doc, indent=None, separators=(" ,", ": "), ensure_ascii=True))      # \\u escapes + surrogate pair
    open(b, "w", encoding="utf-8").write("\n\t " + json.dumps(doc, indent=7, ensure_ascii=False) + " \n")
    order, m, w, banners = smash_mod.load_sketches([a, b], ksize=21, algo="histosketch")
    assert m.tolist() == [mins, mins] and w.tolist() == [[1e-3, -250.0, 0.1, 17.0]] * 2
    assert banners == [doc["banner_label"]] * 2
    _, m31, w31, _ = smash_mod.load_sketches([a, b], ksize=31)
    assert w31.tolist() == [[1.0, 2.0, 3.0, 4.0]] * 2
    _, mk, wk, _ = smash_mod.load_sketches([a, b], ksize=21, algo="khf")
    assert mk.tolist() == [mins, mins] and not wk.any()


def test_md5_of_every_tail_length(tmp_path):
    """the library's own MD5 (RFC 1321) over the little-endian mins (helpers.go:156-166): message lengths 0, 8, ... bytes cross
    both padding cases (one or two final blocks)."""
    files = []
    for n in list(range(1, 20)) + [63, 64, 65, 1000]:
        p = str(tmp_path / f"n{n:04d}.json")
        _write(p, np.arange(n, dtype=np.uint64) * 2654435761 % (21 ** 4), np.ones(n))
        files.append(p)
    for p in files:                                             # each with a partner of its own length
        q = p + ".copy.json"
        open(q, "w").write(open(p).read())
        order, mins, _, _ = smash_mod.load_sketches([p, q])
        assert len(order) == 2 and np.array_equal(mins[0], mins[1])


def _expect(files, text, **kw):
    with pytest.raises(HulkError) as e:
        smash_mod.load_sketches(files, **kw)
    assert text in str(e.value), str(e.value)
    return str(e.value)


def test_error_texts_and_their_order(tmp_path):
    rng = np.random.default_rng(3)
    good = []
    for i in range(3):
        p = str(tmp_path / f"g{i}.json")
        _write(p, *_rand_sketch(rng, 16), filename=f"g{i}.fq,")
        good.append(p)
    raw = json.loads(open(good[0]).read())

    def variant(name, edit):
        d = json.loads(json.dumps(raw))
        edit(d)
        p = str(tmp_path / name)
        open(p, "w").write(json.dumps(d, indent=4))
        return p

    bad_md5 = variant("a_badmd5.json", lambda d: d["signatures"][0]["Sketch"].__setitem__("md5sum", "0" * 32))
    msg = _expect(good + [bad_md5], "md5sum mismatch: " + "0" * 32 + " vs. " + raw["signatures"][0]["Sketch"]["md5sum"] + "\n")
    flipped = variant("a_flipped.json", lambda d: d["signatures"][0]["Sketch"]["mins"].__setitem__(3, d["signatures"][0]["Sketch"]["mins"][3] ^ 1))
    _expect(good + [flipped], "md5sum mismatch: " + raw["signatures"][0]["Sketch"]["md5sum"] + " vs. ")
    _expect(good + [variant("a_nomd5.json", lambda d: d["signatures"][0]["Sketch"].__setitem__("md5sum", ""))], "no MD5 was stored for a sketch: g0.fq,\n")
    _expect(good + [variant("a_ver.json", lambda d: d.__setitem__("version", "0.9.9"))], "the loaded sketch was created with a different version of HULK: 0.9.9\n")
    p = variant("a_cls.json", lambda d: d.__setitem__("class", "sourmash"))
    _expect(good + [p], f"JSON not created by HULK: {p}\n")
    p = variant("a_nosig.json", lambda d: d.__setitem__("signatures", []))
    _expect(good + [p], f"no signatures found in supplied file: {p}\n")
    _expect(good + [variant("a_algo.json", lambda d: d["signatures"][0].__setitem__("Algorithm", "hyperloglog"))], "unknown sketching algorithm: hyperloglog")
    trunc = str(tmp_path / "a_trunc.json")
    open(trunc, "w").write(open(good[0]).read()[:-40])
    _expect(good + [trunc], "malformed sketch file")
    _expect(good + [str(tmp_path / "missing.json")], "No such file")
    deep = str(tmp_path / "a_deep.json")                          # nesting deeper than encoding/json follows: an error, not a stack overflow
    open(deep, "w").write('{"x": ' + "[" * 200_000 + "]" * 200_000 + ', ' + open(good[0]).read().lstrip()[1:])
    _expect(good + [deep], "exceeded max depth")
    # FindSketch (sketchio.go:198-257): raised after every file has loaded, for the first file in sorted order that has one
    _expect(good, "specified k-mer size (31) not found in the supplied sketch file: g0.fq,\n", ksize=31)
    dup = variant("a_dup.json", lambda d: d["signatures"].append(d["signatures"][0]))
    _expect(good + [dup], "found 2 possible duplicate sketches in the supplied sketch file: g0.fq,\n")
    _expect(good, "no sketches were produced using the khf algorithm in file: g0.fq,\n", algo="khf")
    # a load failure comes first even when a file that sorts in front of it would fail FindSketch (cmd/smash.go:165-180)
    k31 = str(tmp_path / "0_k31.json")
    _write(k31, *_rand_sketch(rng, 16), k=31, filename="k31.fq,")
    _expect([k31] + good + [bad_md5], "md5sum mismatch")
    _expect([k31] + good, "specified k-mer size (21) not found in the supplied sketch file: k31.fq,\n")
    # fewer than two sketches (cmd/smash.go:175-177), also when one path is given twice
    _expect([good[0]], "1 sketches found in the supplied directory, HULK needs at least 2 to smash!\n")
    _expect([good[0], good[0]], "1 sketches found in the supplied directory")
    # GetDistance's length check (sketchio.go:274-277): against the first file in sorted order
    longer = str(tmp_path / "z_long.json")
    _write(longer, *_rand_sketch(rng, 20))
    _expect(good + [longer], "sketch length mismatch: 16 vs 20\n")
    assert msg.endswith("\n")


def test_python_and_native_loader_agree_on_the_khf_layout(tmp_path):
    """--khf signatures (src/minhash/khf.go:12-16: ksize, md5sum, mins, num): no weights, loadable under algo = khf."""
    files = []
    for i in range(2):
        d = HULKdata()
        d.filename = "r.fq,"
        khf = KHFSketch(21, 8, np.full(8, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
        khf.algorithm = "khf"
        d.add(khf)
        p = str(tmp_path / f"k{i}.json")
        d.write_json(p)
        files.append(p)
    _, mins, weights, _ = smash_mod.load_sketches(files, algo="khf")
    assert (mins == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and not weights.any()
