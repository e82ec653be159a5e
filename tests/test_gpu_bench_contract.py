"""bench.py contract on the GPU box: ONE JSON line on stdout with the required keys; the sharded-step path through RCCL
at world size 1; and `bench.py --gpus 2` END TO END at world 2 — self-spawn, the pre-warm agreement, the other modes, value_c4
and the JSON relay — with both ranks on this box's one GPU over the library's host transport (HULK_BENCH_TRANSPORT=gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def _run(extra, env_extra=None, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HULK_BENCH_PREWARM_S="0")   # (no need to warm the GPU for a contract test)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
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
    # algorithmic bytes = SURVEY.md §8(d)'s per-read figure (L + 8) x the reads of one launch; nothing else priced in
    assert r["alg_bytes_per_launch"] == a["config"]["reads_per_rank_step"] * (150 + 8) and r["traffic"] is None
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    # the line names the longest single kernel from the durations it measured itself
    assert r["longest_kernel"] == ("k_minimizer_fast" if r["avg_launch_us"] >= a["k_jump_bin"]["avg_launch_us"] else "k_jump_bin")
    assert a["k_jump_left"]["avg_launch_us"] < a["k_jump_bin"]["avg_launch_us"]
    assert a["value"] > 1e7 and abs(a["ms_per_step"] * a["value"] / 1e3 - a["config"]["reads_per_step"]) < 1.0
    assert a["rccl_ranks"] == 0 and a["collective"] is None
    # C2 exactly as BASELINE states it: 10 M reads on a fresh context; below the steady-state rate, above 1e8
    assert a["cold_reads"] == 10_000_000 and 1e8 < a["value_cold"] < 1.2 * a["value"]
    assert a["value_unpruned"] is not None and a["value_unpruned"] <= 1.2 * a["value"]
    # end to end from a FASTQ file, plain and .gz: host-bound, far below the kernel-path rate, same sketch both ways
    e2e = a["e2e"]
    # (one member inflated by several threads and bgzip'd members side by side: both well above one inflate thread's ~5e6, neither
    # ordered against the other nor — on a noisy host — strictly against the plain file)
    for c in ("plain", "gz", "bgzf"):
        assert 1e5 < e2e[c]["value"] < a["value"] and e2e[c]["sketch_md5"] == e2e["plain"]["sketch_md5"]
        assert len(e2e[c]["seconds_all_runs"]) == 4 and min(e2e[c]["seconds_all_runs"]) == e2e[c]["seconds"]
        assert e2e[c]["parse_only_reads_per_s"] > 1e5
    assert e2e["gz"]["value"] < 1.5 * e2e["plain"]["value"] and e2e["bgzf"]["value"] < 1.5 * e2e["plain"]["value"]
    # the sharded step at world size 1: RCCL communicator, exchange inside the library, same sketch
    b = _run(["--no-cpu-baseline", "--no-cold", "--force-collective"], {"HULK_BENCH_C4_READS_PER_RANK": "5000000"})
    assert b["sketch_md5"] == a["sketch_md5"]
    assert b["rccl_ranks"] == 1                         # dist.get_world_size() on the nccl (= RCCL) backend
    cs = b["collective"]["timed_pass"]
    assert "RCCL" in b["collective"]["transport"] and cs["steps_full"] >= 1 and cs["steps_delta"] >= 1
    assert [o["mode"] for o in b["other_scaling"]] == ["sliced-strong", "sliced-weak"]
    for o in b["other_scaling"]:                        # at one rank every mode is the same work and the same sketch
        assert o["reads_per_rank_step"] == b["config"]["reads_per_rank_step"] and 0.5 * b["value"] < o["value"] < 2.0 * b["value"]
        assert o["sketch_md5"] == a["sketch_md5"]
    assert "other_scaling" not in a and "value_c4" not in a
    assert b["c4_reads"] == 5_000_000 and b["c4_exchange"]["steps_delta"] >= 1 and 1e8 < b["value_c4"] < 1.2 * b["value"]
    # SURVEY.md 8(e)'s slice split as the headline: same stream, same interval, same sketch
    f = _run(["--no-cpu-baseline", "--single-pass", "--mode", "sliced-strong", "--force-collective"])
    assert f["sketch_md5"] == a["sketch_md5"] and "slice" in f["config"]["split"] and "whole" in a["config"]["split"]
    assert f["scaling"] == "strong" and "all-reduce" in f["collective"]["per_step"]
    # the real unpruned switch: the timed pass itself reads the whole table for every interval, same sketch
    c = _run(["--no-cpu-baseline", "--no-cold", "--no-e2e", "--no-prune"])
    assert c["sketch_md5"] == a["sketch_md5"]
    sc, sa = c["roofline_cws_scan"], a["roofline_cws_scan"]
    assert sc["tiles_read_per_launch"] == sc["tiles_covered_per_launch"] > 0
    assert sa["tiles_read_per_launch"] < sc["tiles_read_per_launch"]
    assert sc["avg_launch_us"] > sa["avg_launch_us"]
    # variant: 1 % of the reads carry an N (they leave the fast minimizer kernel): measured -2 %, must stay within 15 % here
    e = _run(["--no-cpu-baseline", "--no-cold", "--no-e2e", "--single-pass", "--n-frac", "0.01"])
    assert e["sketch_md5"] != a["sketch_md5"] and e["value"] > 0.85 * a["value"]
    assert "VARIANT" in e["config"]["workload"]
    # one rank's share of an 8-rank step without peers (projection aid): labelled, never a headline
    g = _run(["--no-cpu-baseline", "--single-pass", "--loopback", "8"])
    assert "LOOPBACK" in g["config"]["workload"] and g["config"]["reads_per_step"] == 8 * g["config"]["reads_per_rank_step"]


def test_bench_says_so_when_rccl_cannot_be_bound():
    """hulk_comm_init failing (here: HULK_RCCL_LIB names a file that is not there) must not leave a scaling run without a
    line: every rank learns of it, the run goes over the library's host transport on a gloo group and the line says so."""
    a = _run(["--no-cpu-baseline", "--single-pass", "--no-cold", "--no-e2e", "--no-c4", "--force-collective"], {"HULK_RCCL_LIB": "/nonexistent/librccl.so.1"})
    t = a["collective"]["transport"]
    assert "host transport over gloo" in t and "RCCL unavailable" in t and "librccl" in t
    cs = a["collective"]["timed_pass"]
    assert cs["steps_full"] >= 1 and cs["steps_delta"] >= 1
    b = _run(["--no-cpu-baseline", "--single-pass", "--no-cold", "--no-e2e"])
    assert a["sketch_md5"] == b["sketch_md5"]


def test_bench_world_two_end_to_end_on_one_gpu():
    """`python bench.py --gpus 2` the way the driver invokes it (no launcher), both ranks on GPU 0 over the host transport:
    everything bench.py does at N > 1 runs — self_spawn, the pre-warm agreement, all three modes, value_c4 with its ragged
    last step, the JSON relay — and the sharded sketch is the single-GPU sketch of the same 2 x longer stream."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(HULK_BENCH_TRANSPORT="gloo", HULK_BENCH_PREWARM_S="0.3", HULK_BENCH_C4_READS_PER_RANK="4100000")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["mode"] == "sharded"
    assert out["config"]["reads_per_step"] == 2 * out["config"]["reads_per_rank_step"] == 3_200_000
    assert abs(out["ms_per_step"] * out["value"] / 1e3 - out["config"]["reads_per_step"]) < 1.0
    cs = out["collective"]["timed_pass"]
    assert cs["steps_full"] >= 1 and cs["steps_delta"] >= 1 and cs["bytes_received"] > 0
    assert [o["mode"] for o in out["other_scaling"]] == ["sliced-strong", "sliced-weak"]
    # value_c4: 2 x 4.1 M reads = 82 intervals = 2 whole steps of 32 + a ragged one of 18 (rank 0: 16, rank 1: 2)
    assert out["c4_reads"] == 8_200_000 and out["c4_steps"] == 3 and out["value_c4"] > 1e6      # (a rate over gloo on one GPU: only that it ran)
    # the same global stream on ONE rank (4 steps of 32 intervals = 8 plain steps of 16): the same sketch
    one = _run(["--no-cpu-baseline", "--no-cold", "--no-e2e", "--single-pass", "--steps", "6", "--warmup", "2"])
    assert one["config"]["total_reads"] == 6 * 1_600_000
    assert out["sketch_md5"] == one["sketch_md5"]
    strong = [o for o in out["other_scaling"] if o["mode"] == "sliced-strong"][0]
    half = _run(["--no-cpu-baseline", "--no-cold", "--no-e2e", "--single-pass", "--steps", "3", "--warmup", "1"])
    assert strong["sketch_md5"] == half["sketch_md5"]


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
        assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
