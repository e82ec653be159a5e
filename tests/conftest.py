import gzip
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# torch ships its own HIP runtime: it has to be loaded before libhulkhip.so pulls in /opt/rocm's, or
# torch.cuda finds no device later in the same process (some GPU tests use torch for device buffers)
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pack_reads(seqs):
    """list of bytes -> (bases uint8, offsets uint64)"""
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return bases, offsets


def fixture_reads():
    """The reference's own test input (testing/test-reads-small.fq.gz): 1000 x 100 bp."""
    with gzip.open(os.path.join(GOLDEN, "test-reads-small.fq.gz")) as fh:
        lines = [l.rstrip(b"\n") for l in fh]
    lines = [l for l in lines if l]          # the reference skips empty lines (sketch.go:50,72)
    return [lines[i] for i in range(1, len(lines), 4)]


@pytest.fixture(scope="session")
def fq_reads():
    return fixture_reads()
