#!/usr/bin/env python3
"""What ONE rank of an N-rank strong-scaling run (bench.py --scaling strong) computes per step, timed on
one GPU without the collective: rank 0's slice of every interval (I/N reads), its slot shard (S/N), the replicated
count-min.  step time x N ranks is the compute-only bound of the N-GPU rate; the all-reduce (16 x k^4 uint32 per step)
comes on top where it is not hidden.  Not a benchmark line: a planning aid for the small-shard regime.
usage: shard_projection.py [--worlds 1,2,4,8] [--steps 30]"""
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
from hulk_amd.distributed import GpuEngine, interval_slice, slot_shard

K, W, S, INTERVAL, BATCH, READ_LEN = 21, 9, 512, 100_000, 16, 150


def run(world, steps, warmup=3, split="slice"):
    dev = "cuda:0"
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    os.environ["HULK_BATCH"] = str(BATCH)
    per = interval_slice("strong", 0, INTERVAL, 0, world)[1]
    if split == "interval":      # rank 0 bins the first BATCH / world WHOLE intervals of every batch (spectra 0 .. BATCH/world - 1)
        per = INTERVAL
    sb, sc = slot_shard(S, 0, world)
    n_buf = min(steps + warmup, 12)
    bufs = []
    for s_ in range(n_buf):
        parts = []
        for t in range(BATCH if split == "slice" else BATCH // world):
            first, cnt = interval_slice("strong", s_ * BATCH + t, INTERVAL, 0, world) if split == "slice" else ((s_ * BATCH + t) * INTERVAL, INTERVAL)
            b, _ = synth.reads_torch(first, cnt, READ_LEN, device=dev)
            parts.append(b[:cnt * READ_LEN])
        bufs.append(torch.cat(parts + [torch.zeros(16, dtype=torch.uint8, device=dev)]))
    n_step = per * BATCH if split == "slice" else INTERVAL * (BATCH // world)
    offsets = torch.arange(n_step + 1, dtype=torch.int64, device=dev) * READ_LEN
    sk = hulk_amd.GpuSketcher(K, W, S, interval=0, decay_ratio=1.0, device=0, slot_begin=sb, slot_count=sc,
                              stream=stream.cuda_stream)

    eng = GpuEngine(sk, dev, n_spectra=BATCH)
    own = BATCH // world

    def step(t):
        b = bufs[t % n_buf]
        sk.bin_reads_device(b.data_ptr(), offsets.data_ptr(), n_step, READ_LEN, b.numel(), reads_per_spectrum=per)
        if split == "interval" and world > 1:
            # what the gather would bring: the other ranks' spectra (copies of this rank's first one stand in for them), so
            # that the flush sees BATCH non-empty spectra as it does in an N-rank run
            h = eng.histogram_tensor().view(BATCH, -1)
            h[own:] = h[0]                                    # (~10 us on the work stream; the real gather runs on its own)
        sk.flush_batch(BATCH)
    for t in range(warmup):
        step(t)
    sk.synchronize(); torch.cuda.synchronize()
    sk.set_profiling(True)
    t0 = time.perf_counter()
    for t in range(warmup, warmup + steps):
        step(t)
    sk.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = {k: sk.get_profile(k) for k in ("k_minimizer_fast", "k_jump_bin")}
    sk.close()
    return {"world": world, "split": split, "reads_per_rank_step": n_step, "slots": sc, "ms_per_step": dt * 1e3,
            "k1a_us": prof["k_minimizer_fast"][1] * 1e3 / max(prof["k_minimizer_fast"][0], 1),
            "k1b_us": prof["k_jump_bin"][1] * 1e3 / max(prof["k_jump_bin"][0], 1),
            "compute_only_reads_per_s": INTERVAL * BATCH / dt}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--split", choices=("slice", "interval"), default="slice",
                    help="slice: a rank bins its 1/N of every interval (SURVEY.md 8e); interval: whole intervals, BATCH/N per batch")
    a = ap.parse_args()
    base = None
    for w in [int(x) for x in a.worlds.split(",")]:
        r = run(w, a.steps, split=a.split)
        if base is None:
            base = r["compute_only_reads_per_s"]
        r["speedup_bound"] = r["compute_only_reads_per_s"] / base
        print(json.dumps(r), flush=True)
