#!/usr/bin/env python3
"""What ONE rank of a G-rank run of hulk_step_sharded computes per step, timed on one GPU without peers
(hulk_comm_init_loopback: the other ranks' contributions to the all-gather are copies of this rank's own): rank 0's 16
whole intervals of the step (1.6 M reads), its slot shard (S/G), the replicated count-min upkeep of all G x 16 intervals.
G x 1.6 M reads / step time is the compute-only bound of the G-GPU rate; the all-gather (G x 0.9 MB of count-min
increments per step in steady state) runs on the flush stream beside it.  Not a benchmark line: a planning aid.
usage: shard_projection.py [--worlds 1,2,4,8] [--steps 30] [--mode sharded|sliced]
  sliced = SURVEY.md 8(e) to the letter (rank 0's 1/G slice of each of 16 intervals, the all-reduce an identity here)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hulk_amd
from hulk_amd import synth
from hulk_amd.distributed import interval_slice, slot_shard, step_share

K, W, S, INTERVAL, BATCH, READ_LEN = 21, 9, 512, 100_000, 16, 150


def run(world, steps, warmup=3, mode="sharded"):
    dev = "cuda:0"
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sb, sc = slot_shard(S, 0, world)
    n_buf = min(steps + warmup, 12)
    bufs = []
    per = INTERVAL if mode == "sharded" else interval_slice("strong", 0, INTERVAL, 0, world)[1]
    for s_ in range(n_buf):
        parts = []
        if mode == "sharded":
            first, cnt, _ = step_share(s_, BATCH, INTERVAL, 0, world)
            b, _ = synth.reads_torch(first, cnt, READ_LEN, device=dev)
            parts.append(b[:cnt * READ_LEN])
        else:
            for t in range(BATCH):
                first, cnt = interval_slice("strong", s_ * BATCH + t, INTERVAL, 0, world)
                b, _ = synth.reads_torch(first, cnt, READ_LEN, device=dev)
                parts.append(b[:cnt * READ_LEN])
        bufs.append(torch.cat(parts + [torch.zeros(16, dtype=torch.uint8, device=dev)]))
    n_step = per * BATCH
    offsets = torch.arange(n_step + 1, dtype=torch.int64, device=dev) * READ_LEN
    sk = hulk_amd.GpuSketcher(K, W, S, interval=INTERVAL if mode == "sharded" else 0, decay_ratio=1.0, device=0,
                              slot_begin=sb, slot_count=sc, batch=BATCH)
    sk.comm_init_loopback(0, world)

    def step(t):
        b = bufs[t % n_buf]
        if mode == "sharded":
            sk.step_sharded(b.data_ptr(), offsets.data_ptr(), n_step, READ_LEN, b.numel(), world * BATCH)
        else:
            sk.step_sliced(b.data_ptr(), offsets.data_ptr(), n_step, READ_LEN, b.numel(), per, BATCH)
    for t in range(warmup):
        step(t)
    sk.synchronize(); torch.cuda.synchronize()
    sk.set_profiling(True)
    t0 = time.perf_counter()
    for t in range(warmup, warmup + steps):
        step(t)
    sk.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = {k: sk.get_profile(k) for k in ("k_minimizer_fast", "k_jump_bin", "k_jump_left")}
    stats = sk.comm_stats()
    sk.close()
    global_reads = (world if mode == "sharded" else 1) * INTERVAL * BATCH
    return {"world": world, "mode": mode, "reads_per_rank_step": n_step, "global_reads_per_step": global_reads, "slots": sc,
            "ms_per_step": dt * 1e3,
            "k1a_us": prof["k_minimizer_fast"][1] * 1e3 / max(prof["k_minimizer_fast"][0], 1),
            "k1b_us": (prof["k_jump_bin"][1] + prof["k_jump_left"][1]) * 1e3 / max(prof["k_jump_bin"][0], 1),
            "exchange": stats, "compute_only_reads_per_s": global_reads / dt}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--mode", choices=("sharded", "sliced"), default="sharded")
    a = ap.parse_args()
    run(1, 8)              # (discarded: the first context of a process runs well below the ones after it)
    base = None
    for w in [int(x) for x in a.worlds.split(",")]:
        r = run(w, a.steps, mode=a.mode)
        if base is None and w == 1:
            base = r["compute_only_reads_per_s"]
        if base:
            r["bound_vs_one_gpu"] = r["compute_only_reads_per_s"] / base
        print(json.dumps(r), flush=True)
