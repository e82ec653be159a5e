#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the CPU oracle (the C restatement, NOT Go output — the
reference cannot be built in this image; provenance is recorded in each file's banner_label).

  c1_fixture_k21_s256.json : BASELINE config C1 — testing/test-reads-small.fq.gz, k=21, w=9,
                             sketchSize=256, no interval (`hulk sketch -f ... -k 21 --sketchSize 256`)
  c1_fixture_k15_s64_drift.json : same reads, k=15, sketchSize=64, interval=250, decayRatio=0.05
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import fixture_reads, GOLDEN
from oracle import pyorc
from hulk_amd.sketcher import HistoSketch
from hulk_amd.sketchio import HULKdata

reads = fixture_reads()
for name, k, S, interval, decay in (("c1_fixture_k21_s256", 21, 256, 0, 1.0),
                                    ("c1_fixture_k15_s64_drift", 15, 64, 250, 0.05)):
    o = pyorc.Sketcher(k, 9, S, 0, decay, interval)
    for r in reads:
        o.add_read(r)
    o.finish()
    mins, weights = o.sketch()
    d = HULKdata()
    d.add(HistoSketch(k, mins, weights, o.B, decay != 1.0))
    d.filename = "testing/test-reads-small.fq.gz,"
    d.banner_label = "golden: CPU restatement (oracle/hulk_oracle.c), not Go output"
    d.write_json(os.path.join(GOLDEN, name + ".json"))
    print(name, "written;", o.counters())
    o.close()
