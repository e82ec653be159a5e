"""Pins the oracle against REAL Go output when a maintainer has produced it with tools/go/ (see
tools/go/README.md).  Without those files (this repository cannot build Go: no toolchain, no network)
the tests skip and CWS parity with Go-produced sketches stays "unpinned" (DESIGN.md §6)."""
import glob
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import pyorc


def _cws_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "go_cws_s*_k*.bin")))


@pytest.mark.skipif(not _cws_files(), reason="no tests/golden/go_cws_s*_k*.bin (tools/go/dump_cws not run)")
@pytest.mark.parametrize("path", _cws_files() or ["-"])
def test_cws_tables_against_go(path):
    m = re.search(r"go_cws_s(\d+)_k(\d+)\.bin$", path)
    S, k = int(m.group(1)), int(m.group(2))
    B = k ** 4
    raw = np.fromfile(path, dtype="<f8")
    assert raw.size == 3 * S * B
    r, c, b = raw.reshape(3, S, B)
    orr, oc, ob = pyorc.cws_tables(S, B)
    assert np.array_equal(orr, r) and np.array_equal(oc, c) and np.array_equal(ob, b)


def _golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "go_c1_k*_s*.json")))


@pytest.mark.skipif(not _golden_files(), reason="no tests/golden/go_c1_k*_s*.json (tools/go/dump_golden not run)")
@pytest.mark.parametrize("path", _golden_files() or ["-"])
def test_fixture_sketch_against_go(path, fq_reads):
    m = re.search(r"go_c1_k(\d+)_s(\d+)\.json$", path)
    k, S = int(m.group(1)), int(m.group(2))
    doc = json.load(open(path))
    o = pyorc.Sketcher(k, 9, S)
    for rd in fq_reads:
        o.add_read(rd)
    o.finish()
    mins, weights = o.sketch()
    assert o.counters()["n_minimizers"] == doc["n_minimizers"]
    assert np.array_equal(mins, np.array(doc["mins"], dtype=np.uint64))
    assert np.allclose(weights, np.array(doc["weights"]), rtol=1e-12, atol=0)
    o.close()
