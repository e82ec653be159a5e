#!/usr/bin/env python3
"""Hunt for the intermittent failure of the multi-rank bench run (commit 51a2aa5: "one unexplained failure in ~60 runs" of
tests/test_gpu_bench_contract.py::test_bench_world_two_end_to_end_on_one_gpu, its text lost): the SAME invocation the test makes
— `python bench.py --gpus W --steps 3 --warmup 1 --no-cpu-baseline` with all ranks on GPU 0 — N times, JOBS at a time (several
runs side by side load the box the way a busy node would), every run checked the way the test checks it: exit status 0, ONE
JSON line, no `<leg>_error` key, the sharded sketch's MD5 equal in every run, hulk_get_comm_health all zero.  The complete
stdout + stderr of every failing run is kept under gpurun_out/flake/ and the tally is printed (-> profiles/r06_flake.txt).

usage: gpu_flake_hunt2.py N [--world 2] [--transport gloo|fakerccl] [--jobs 3] [--seconds LIMIT]"""
import argparse
import concurrent.futures as cf
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("n", type=int)
ap.add_argument("--world", type=int, default=2)
ap.add_argument("--transport", default="gloo")
ap.add_argument("--jobs", type=int, default=3)
ap.add_argument("--seconds", type=float, default=0.0, help="stop starting new runs after this many seconds")
ap.add_argument("--c4", default=None, help="HULK_BENCH_C4_READS_PER_RANK (default: the contract test's value for this world)")
a = ap.parse_args()
out_dir = os.path.join(ROOT, "gpurun_out", "flake")
os.makedirs(out_dir, exist_ok=True)
c4 = a.c4 or ("4100000" if a.world == 2 else "1850000")
t_start = time.time()


def one(i):
    if a.seconds and time.time() - t_start > a.seconds:
        return i, "skipped", None, 0.0
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HULK_BENCH_TRANSPORT=a.transport, HULK_BENCH_PREWARM_S="0.3" if a.world == 2 else "0", HULK_BENCH_C4_READS_PER_RANK=c4,
               HULK_BENCH_LONG_STEPS="6" if a.world == 2 else "0", HULK_BENCH_LEG_TIMEOUT_S="240")
    t0 = time.time()
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(a.world), "--steps", "3" if a.world == 2 else "2",
                            "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
        rc, so, se = p.returncode, p.stdout, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, so, se = -999, (e.stdout or b"").decode("utf-8", "replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), \
            (e.stderr or b"").decode("utf-8", "replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
    dt = time.time() - t0
    why, md5 = None, None
    lines = [l for l in so.splitlines() if l.startswith("{")]
    if rc != 0:
        why = f"exit status {rc}"
    elif len(lines) != 1:
        why = f"{len(lines)} JSON lines"
    else:
        d = json.loads(lines[0])
        md5 = (d.get("sketch_md5"), d.get("c4_sketch_md5"), tuple(o.get("sketch_md5") for o in d.get("other_scaling", [])))
        errs = {k: d[k] for k in d if k.endswith("_error")}
        cs = (d.get("collective") or {}).get("timed_pass") or {}
        health = [(o.get("mode"), (o.get("exchange") or {}).get("headers_refetched"), (o.get("exchange") or {}).get("void_blocks")) for o in d.get("other_scaling", [])]
        if errs:
            why = f"leg errors: {errs}"
        elif cs.get("headers_refetched") or cs.get("void_blocks") or (d.get("c4_exchange") or {}).get("void_blocks") or (d.get("c4_exchange") or {}).get("headers_refetched"):
            why = f"comm health not zero: timed pass {cs}, c4 {d.get('c4_exchange')}, others {health}"
        elif d.get("n_gpus") != a.world:
            why = f"n_gpus {d.get('n_gpus')}"
    if why:
        with open(os.path.join(out_dir, f"fail_{a.transport}_w{a.world}_{i:04d}.txt"), "w") as fh:
            fh.write(f"run {i}: {why}\nreturncode {rc}, {dt:.1f} s\n---- stdout\n{so[-60000:]}\n---- stderr\n{se[-120000:]}\n")
    return i, why, md5, dt


fails, md5s, done, secs = [], {}, 0, []
with cf.ThreadPoolExecutor(max_workers=a.jobs) as ex:
    for i, why, md5, dt in ex.map(one, range(a.n)):
        if why == "skipped":
            continue
        done += 1; secs.append(dt)
        if why:
            fails.append((i, why))
            print(f"run {i}: FAIL {why[:300]} ({dt:.1f} s)", flush=True)
        else:
            md5s[md5] = md5s.get(md5, 0) + 1
            if done % 20 == 0:
                print(f"... {done} runs, {len(fails)} failures, {len(md5s)} distinct sketch tuples", flush=True)
print(f"flake hunt: bench.py --gpus {a.world} over {a.transport}, {a.jobs} at a time: {done} runs in {time.time() - t_start:.0f} s "
      f"(median {sorted(secs)[len(secs) // 2] if secs else 0:.1f} s per run), {len(fails)} failed, {len(md5s)} distinct "
      f"(sharded, c4, other modes) sketch-MD5 tuples among the {done - len(fails)} that passed")
for i, why in fails:
    print(f"  run {i}: {why[:1000]}")
if len(md5s) > 1:
    print("  DISTINCT SKETCHES:", md5s)
sys.exit(1 if fails or len(md5s) > 1 else 0)
