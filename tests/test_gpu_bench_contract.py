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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HULK_BENCH_PREWARM_S="0")   # (no need to warm the GPU for a contract test)
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
    assert a["unit"] == "reads/s" and a["higher_is_better"] is True and a["scaling"] == "strong"
    assert "workload" in a["config"] and "model" not in a["config"]
    r = a["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # algorithmic bytes = SURVEY.md §8(d)'s per-read figure (L + 8) x the reads of one launch; nothing else priced in
    assert r["alg_bytes_per_launch"] == a["config"]["reads_per_rank_step"] * (150 + 8) and r["traffic"] is None
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert a["value"] > 1e7 and abs(a["ms_per_step"] * a["value"] / 1e3 - a["config"]["reads_per_step"]) < 1.0
    assert a["rccl_ranks"] == 0
    # C2 exactly as BASELINE states it: 10 M reads on a fresh context; below the steady-state rate, above 1e8
    assert a["cold_reads"] == 10_000_000 and 1e8 < a["value_cold"] < 1.2 * a["value"]
    assert a["value_unpruned"] is not None and a["value_unpruned"] <= 1.2 * a["value"]
    b = _run(["--no-cpu-baseline", "--no-cold", "--force-collective"])
    assert b["sketch_md5"] == a["sketch_md5"]           # all-reduce over one rank is the identity
    assert b["rccl_ranks"] == 1                         # dist.get_world_size() on the nccl (= RCCL) backend
    assert "all_gather" in b["collective"] and a["collective"] is None     # whole intervals per rank: the exchange is a gather
    # with a collective the same steps are also timed under the other scaling rule (at one rank: the same work)
    assert [o["mode"] for o in b["other_scaling"]] == ["strong", "weak"]      # the headline splits a batch by whole intervals
    for o in b["other_scaling"]:
        assert o["reads_per_rank_step"] == b["config"]["reads_per_rank_step"] and 0.5 * b["value"] < o["value"] < 2.0 * b["value"]
    assert "other_scaling" not in a
    # the slice split of SURVEY.md 8(e) as the headline: same stream, same interval, same sketch
    f = _run(["--no-cpu-baseline", "--no-cold", "--single-pass", "--split", "slice", "--force-collective"])
    assert f["sketch_md5"] == a["sketch_md5"] and "slice" in f["config"]["split"] and "whole" in a["config"]["split"]
    assert "all_reduce" in f["collective"]
    # the real unpruned switch: the timed pass itself reads the whole table for every interval, same sketch
    c = _run(["--no-cpu-baseline", "--no-cold", "--no-prune"])
    assert c["sketch_md5"] == a["sketch_md5"]
    sc, sa = c["roofline_cws_scan"], a["roofline_cws_scan"]
    assert sc["tiles_read_per_launch"] == sc["tiles_covered_per_launch"] > 0
    assert sa["tiles_read_per_launch"] < sc["tiles_read_per_launch"]
    assert sc["avg_launch_us"] > sa["avg_launch_us"]
    # variant: 1 % of the reads carry an N (they leave the fast minimizer kernel): measured -2 %, must stay within 15 % here
    e = _run(["--no-cpu-baseline", "--no-cold", "--single-pass", "--n-frac", "0.01"])
    assert e["sketch_md5"] != a["sketch_md5"] and e["value"] > 0.85 * a["value"]
    assert "VARIANT" in e["config"]["workload"]
    # weak scaling is identical to strong at one rank
    d = _run(["--no-cpu-baseline", "--no-cold", "--single-pass", "--scaling", "weak"])
    assert d["scaling"] == "weak" and d["sketch_md5"] == a["sketch_md5"]


def test_bench_gpus_2_spawns_or_refuses():
    """Invoked the way the driver invokes it (`python bench.py --gpus 2`, no launcher): on a one-GPU box it must refuse
    with a non-zero exit — never print an n_gpus: 1 line; with two GPUs it spawns two RCCL ranks."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and "GPU(s) visible" in p.stderr and "{" not in p.stdout
    else:
        assert p.returncode == 0, p.stderr[-2000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "strong"
