"""bench.py contract on the GPU box: ONE JSON line on stdout with the required keys, with and without
the collective path (RCCL at world size 1), same sketch either way."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def _run(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_json_contract_and_collective_path():
    a = _run(["--no-cpu-baseline"])
    for k in REQUIRED:
        assert k in a, k
    assert a["n_gpus"] == 1 and a["steps"] == 3 and a["warmup"] == 1 and a["vs_baseline"] is None
    assert a["unit"] == "reads/s" and a["higher_is_better"] is True and a["scaling"] == "weak"
    assert "workload" in a["config"] and "model" not in a["config"]
    r = a["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert a["value"] > 1e7 and abs(a["ms_per_step"] * a["value"] / 1e3 - a["config"]["reads_per_step"]) < 1.0
    b = _run(["--no-cpu-baseline", "--force-collective"])
    assert b["sketch_md5"] == a["sketch_md5"]           # all-reduce over one rank is the identity
