"""The hardware property the count-min replay rests on, and its guard.

k_cms_freq / k_cmsd_freq (hulk_amd/csrc/hulk_countmin.hip) take the value a returning LDS atomic add hands back as "the counter
before this bin" — the reference's CountMinSketch.Add in bin order (src/countmin/countmin.go:103-138) — which is exact only if
the LDS applies the lanes of one instruction that hit one address in ascending lane order.  hulk_create checks that on the
device (once per process and device: all-equal, paired and pseudo-random address patterns, ds_add_rtn_u64 and _f64) and, where
it does not hold — or with HULK_FLAG_CMS_CHAIN — runs the chain-form kernels, which take the order from a static table and
register exchanges instead.  Here: the probe passes on this chip; the fallback is bit-identical to the default (counters,
estimates-dependent sketch) and matches the oracle; a probe that reports a violation (forced in the profiling build) selects
the fallback and says so."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import pack_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reads(seed, n, L=150):
    rng = np.random.default_rng(seed)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [bytes(a[rng.integers(0, 4, size=L)]) for _ in range(n)]


def _run(k, S, interval, decay, batch, flags, seqs):
    import hulk_amd
    g = hulk_amd.GpuSketcher(k, 9, S, interval=interval, decay_ratio=decay, batch=batch, flags=flags)
    chk = g.device_checks()
    bases, offsets = pack_reads(seqs)
    g.add_reads(bases, offsets)
    g.finish()
    m, w = g.sketch()
    out = (m, w, g.cms(), g.counters(), chk)
    g.close()
    return out


def test_lds_atomic_order_guard():
    """hulk_create's self-test passes on this device and the default contexts run the returning-atomic kernels; the flag selects
    the chain form."""
    import hulk_amd
    from hulk_amd import _lib
    g = hulk_amd.GpuSketcher(15, 9, 8)
    assert g.device_checks() == {"lds_order_ok": True, "cms_chain_form": False}
    g.close()
    g = hulk_amd.GpuSketcher(15, 9, 8, flags=_lib.HULK_FLAG_CMS_CHAIN)
    assert g.device_checks() == {"lds_order_ok": True, "cms_chain_form": True}
    g.close()


@pytest.mark.parametrize("k,S,interval,decay,batch,n", [
    (15, 32, 2_000, 1.0, 4, 21_300),            # integer counters, several batches, a ragged last one
    (21, 16, 5_000, 1.0, 16, 25_000),           # the default k: 194,481 bins, 3039 chunks of 64
    (31, 4, 5_000, 0.02, 3, 30_000),            # test_k31_concept_drift_against_oracle's configuration: fp64 counters with decay
    (31, 4, 5_000, 0.3, 16, 30_000),
    (13, 8, 700, 0.97, 5, 9_000),               # decay so strong that the base element moves inside a segment
])
def test_forced_chain_fallback_is_bit_identical_and_matches_the_oracle(k, S, interval, decay, batch, n):
    from hulk_amd import _lib
    from oracle import pyorc
    seqs = _reads(1000 * k + int(decay * 100), n)
    a = _run(k, S, interval, decay, batch, 0, seqs)
    b = _run(k, S, interval, decay, batch, _lib.HULK_FLAG_CMS_CHAIN, seqs)
    assert not a[4]["cms_chain_form"] and b[4]["cms_chain_form"]
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))        # weights: same additions in the same order, to the bit
    assert np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))        # count-min counters
    assert a[3] == b[3]
    o = pyorc.Sketcher(k, 9, S, 0, decay, interval)
    bases, offsets = pack_reads(seqs)
    o.add_reads(bases, offsets)
    o.finish()
    om, ow = o.sketch()
    assert np.array_equal(om, b[0])
    assert np.allclose(b[1], ow, rtol=1e-7 if decay != 1.0 else 1e-9, atol=0)
    assert np.allclose(b[2], o.cms(), rtol=1e-9, atol=1e-300)
    o.close()


def test_a_probe_that_reports_a_violation_selects_the_fallback():
    """The checker itself: in the profiling build HULK_LDS_PROBE_SABOTAGE makes one lane's expectation wrong, which is what a
    device with another lane order would look like to it — the context must then run the chain-form kernels, say so on stderr
    and through hulk_get_device_checks, and still produce the default's sketch."""
    if not os.path.exists(os.path.join(ROOT, "hulk_amd", "csrc", "libhulkhip_exp.so")):
        pytest.skip("profiling build (make -C hulk_amd/csrc EXPERIMENTS=1) not present")
    code = (
        "import sys, hashlib, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import torch, hulk_amd\n"
        "from hulk_amd import _lib, synth\n"
        "bases, offsets = synth.reads_numpy(5, 24000, 150)\n"
        "g = hulk_amd.GpuSketcher(31, 9, 4, interval=5000, decay_ratio=0.02, batch=3)\n"
        "chk = g.device_checks()\n"
        "g.add_reads(bases, offsets); g.finish()\n"
        "m, w = g.sketch()\n"
        "print(json.dumps({'chk': chk, 'exp': _lib.is_experiments_build(), 'md5': hashlib.md5(m.tobytes() + w.tobytes() + g.cms().tobytes()).hexdigest()}))\n"
        "g.close()\n")
    import json
    out = {}
    for label, extra in (("default", {}), ("sabotaged", {"HULK_LDS_PROBE_SABOTAGE": "1"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HULK_LIB="exp", **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[label] = (json.loads(r.stdout.strip().splitlines()[-1]), r.stderr)
    d, s_ = out["default"][0], out["sabotaged"][0]
    assert d["exp"] and s_["exp"]
    assert d["chk"] == {"lds_order_ok": True, "cms_chain_form": False}
    assert s_["chk"] == {"lds_order_ok": False, "cms_chain_form": True}
    assert "chain-form kernels" in out["sabotaged"][1] and "chain-form kernels" not in out["default"][1]
    assert d["md5"] == s_["md5"]
