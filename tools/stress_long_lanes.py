#!/usr/bin/env python3
"""The long-sequence path on two work lanes (it returns with its kernels queued; one descriptor / set-table scratch per context): random
k, w, interval, batch size, call boundaries, reads of 1.1-60 kb mixed with short ones, with and without decay — sketch and minimizer count
against the oracle.  usage: stress_long_lanes.py [cases] [seed]     FUZZ_SECONDS stops it early"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import hulk_amd
from oracle import pyorc
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
budget = float(os.environ.get("FUZZ_SECONDS", 0))
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
done = 0
for case in range(n_cases):
    if budget and time.time() - t0 > budget:
        break
    k = int(rng.choice([9, 11, 15, 21]))
    w = int(rng.choice([1, 2, 5, 9, 16]))
    S = int(rng.choice([8, 31, 64]))
    decay = float(rng.choice([1.0, 1.0, 0.02]))
    I = int(rng.choice([1, 3, 8, 16]))
    batch = int(rng.choice([1, 1, 2, 4]))
    n = int(rng.integers(20, 160))
    lens = np.where(rng.random(n) < 0.8, rng.integers(1100, 60_000, size=n), rng.integers(w + k - 1, 400, size=n))
    alph = np.frombuffer([b"ACGT", b"ACGTN", b"ACGTacgtN"][int(rng.integers(0, 3))], dtype=np.uint8)
    seqs = [bytes(alph[rng.integers(0, len(alph), size=int(l))]) for l in lens]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offsets = np.zeros(n + 1, dtype=np.uint64); np.cumsum(lens, out=offsets[1:])
    o = pyorc.Sketcher(k, w, S, 0, decay, I)
    g = hulk_amd.GpuSketcher(k, w, S, I, decay, 0, batch=batch, work_lanes=2)
    desc = f"case {case}: k={k} w={w} S={S} decay={decay} I={I} batch={batch} reads={n}"
    try:
        o.add_reads(bases, offsets); o.finish()
        oerr = None
    except Exception as e:                                     # (e.g. "not used yet": the 1 % rule)
        oerr = str(e)
    try:
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 6)))]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            g.add_reads(bases, offsets[a:b + 1])
        g.finish()
        gerr = None
    except hulk_amd.HulkError as e:
        gerr = e.message
    if oerr or gerr:
        if bool(oerr) != bool(gerr):
            bad += 1; print("MISMATCH", desc, "::", oerr, "|", gerr, flush=True)
    else:
        om, ow = o.sketch(); gm, gw = g.sketch()
        if not (np.array_equal(om, gm) and np.allclose(gw, ow, rtol=1e-7, atol=0) and o.counters()["n_minimizers"] == g.counters()["n_minimizers"]):
            bad += 1; print("MISMATCH", desc, "::", int((om != gm).sum()), "mins differ", flush=True)
    g.close(); o.close()
    done += 1
print(f"{done} cases, {bad} mismatches, {time.time() - t0:.1f} s (seed {seed})")
sys.exit(1 if bad else 0)
