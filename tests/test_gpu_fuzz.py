"""A fixed-seed slice of the randomised GPU-vs-oracle differential test (tools/fuzz_parity.py: random k, w,
sketch size, decay, interval, batch size, alphabets with N / lowercase / odd bytes, length mixes that hit
the short-read, one-wave and long-sequence kernels, random call boundaries).  2100 cases over three other
seeds were clean when this was written."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_slice():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "120", "7"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 mismatches" in p.stdout
