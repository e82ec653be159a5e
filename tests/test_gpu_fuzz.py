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


@pytest.mark.parametrize("full", [False, True])
def test_fuzz_slice_of_the_multi_rank_protocol(full):
    """tools/fuzz_shard.py: G = 1..8 ranks as threads of one process on the one GPU, connected by an in-process exchange
    function over the library's host transport; hulk_step_sharded (spectra / count-min-increment exchange, ragged last steps,
    ranks without reads, drift) and hulk_step_sliced against the oracle.  full: the spectra exchange on every step."""
    env = dict(os.environ)
    if full:
        env["FUZZ_SHARD_FULL"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_shard.py"), "30", "17" if full else "16"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 mismatches" in p.stdout


def test_fuzz_slice_of_the_multi_rank_protocol_with_sixteen_hardware_queues():
    """The same slice with GPU_MAX_HW_QUEUES=16 — the setting under which round 4 met a void exchange header once per ~20
    runs.  The exchange now runs on the flush stream with sealed, self-checking headers: the sketches must match and the
    summary line must carry the header-health counters (hulk_get_comm_health over every rank of every case)."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_shard.py"), "30", "16"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 mismatches" in p.stdout and "void blocks: 0" in p.stdout


def test_fuzz_slice_of_the_device_fastq_parser():
    """tools/fuzz_devparse.py: hulk_sketch_files with the line machine on the GPU (hulk_fastq.hip) and, beside it, on the host
    (HULK_INGEST_HOST_PARSER), against oracle/linepump.py: line soups with empty lines, CR/LF, stray headers, 64 KiB lines,
    several inputs, gzip, blocks of 128 KiB..1 MiB so that records straddle the borders — reads, total length, line count and
    the k-mer spectrum of everything binned, or the reference's error text."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_devparse.py"), "60", "3"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 mismatches" in p.stdout
