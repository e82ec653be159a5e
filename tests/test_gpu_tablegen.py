"""CWS table generation with the math/rand stream produced on the device (k_alfg_jump / k_alfg_fill + the event
list of k_cws_eval) against the host walk of the same stream (HULK_CWS_HOST=1), bit for bit, at a size where
attempts that die on the u1 range test do occur (2e-7 each: ~6 in 3.2e7 attempts), and for a slot shard."""
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
t0 = time.time()
g = hulk_amd.GpuSketcher(%d, 9, %d, slot_begin=%d, slot_count=%d)
dt = time.time() - t0
r, c, b = g.cws_tables()
print(hashlib.sha256(r.tobytes()).hexdigest(), hashlib.sha256(c.tobytes()).hexdigest(), hashlib.sha256(b.tobytes()).hexdigest(), "%%.3f" %% dt)
g.close()
"""


def run(k, S, sb, sc, host):
    env = dict(os.environ)
    env.pop("HULK_CWS_HOST", None)
    if host:
        env["HULK_CWS_HOST"] = "1"
    p = subprocess.run([sys.executable, "-c", CODE % (ROOT, k, S, sb, sc)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout.split()


@pytest.mark.parametrize("k,S,sb,sc", [(21, 64, 0, 0), (21, 48, 16, 24), (13, 7, 0, 0)])
def test_device_stream_equals_host_walk(k, S, sb, sc):
    d = run(k, S, sb, sc, host=False)
    h = run(k, S, sb, sc, host=True)
    assert d[:3] == h[:3]
    print("create: device %s s, host %s s" % (d[3], h[3]))
