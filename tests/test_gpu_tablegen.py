"""CWS table generation with the math/rand stream produced on the device (k_alfg_jump / k_alfg_fill + the event
list of k_cws_eval) against the host walk of the same stream, bit for bit, at a size where attempts that die on the u1
range test do occur (2e-7 each: ~6 in 3.2e7 attempts), and for a slot shard.  The host walk is an experiment switch
(HULK_CWS_HOST=1) and exists in the PROFILING build only (HULK_LIB=exp): the comparator leg loads that build, says so
(hulk_build_info ends in " experiments=1") and reports how the tables were made; the device leg runs the shipping library."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = """
import sys, hashlib, time
sys.path.insert(0, %r)
import hulk_amd
from hulk_amd import _lib
t0 = time.time()
g = hulk_amd.GpuSketcher(%d, 9, %d, slot_begin=%d, slot_count=%d)
dt = time.time() - t0
r, c, b = g.cws_tables()
print(hashlib.sha256(r.tobytes()).hexdigest(), hashlib.sha256(c.tobytes()).hexdigest(), hashlib.sha256(b.tobytes()).hexdigest(), "%%.3f" %% dt, "exp" if _lib.is_experiments_build() else "ship")
g.close()
"""


def run(k, S, sb, sc, host):
    env = dict(os.environ)
    env.pop("HULK_CWS_HOST", None)
    env.pop("HULK_LIB", None)
    if host:
        if not os.path.exists(os.path.join(ROOT, "hulk_amd", "csrc", "libhulkhip_exp.so")):
            pytest.skip("profiling build (make -C hulk_amd/csrc EXPERIMENTS=1) not present")
        env["HULK_CWS_HOST"] = "1"
        env["HULK_LIB"] = "exp"                                 # (the shipping library compiles the switch out: it would run the device path twice)
    p = subprocess.run([sys.executable, "-c", CODE % (ROOT, k, S, sb, sc)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout.split()


@pytest.mark.parametrize("k,S,sb,sc", [(21, 64, 0, 0), (21, 48, 16, 24), (13, 7, 0, 0)])
def test_device_stream_equals_host_walk(k, S, sb, sc):
    d = run(k, S, sb, sc, host=False)
    h = run(k, S, sb, sc, host=True)
    assert d[4] == "ship" and h[4] == "exp"                     # the switch was in effect: the legs are two different generators
    assert d[:3] == h[:3]
    # (host walk: one thread steps through the whole math/rand stream; the device jump-ahead is the faster one at any size that matters)
    print("create: device %s s, host %s s" % (d[3], h[3]))
