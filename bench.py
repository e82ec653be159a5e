#!/usr/bin/env python3
"""bench.py — reads/sec of the HULK `sketch` hot path on MI355X (BASELINE.json metric).

Workload at N=1 (BASELINE configs[1], "C2"): synthetic 150 bp reads, k=21, w=9, sketchSize=512,
interval=100k reads; reads are resident in HBM before the timed region.  One *step* = one batch of
T=16 sketching intervals (1.6 M reads): one launch chain bins the reads of the 16 intervals into 16
k-mer spectra (minimizers -> jump hash -> spectrum), the spectra go through the count-min update,
and ONE pass over the CWS table applies all 16 histosketch updates in interval order —
bit-identical to flushing after every 100k reads (tests/test_gpu_parity.py).  The flush of step n
runs on a second stream under the minimizer kernels of step n+1.  Default K=20 steps = 32 M reads
(C2's 10 M reads = 6.25 steps).

N>1 (one process per GPU, launched by torch.distributed.run): every interval's reads are split
into N contiguous slices, the 16 spectra of a step are merged with ONE RCCL all-reduce, the CWS
update is slot-sharded (hulk_amd/distributed.py).  Per-rank work per step is kept fixed as N
grows (each rank bins 100k reads per interval => the global interval is N x 100k): "weak" scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, W, S, READ_LEN, INTERVAL = 21, 9, 512, 150, 100_000
BATCH = int(os.environ.get("HULK_BENCH_BATCH", "16"))   # sketching intervals per step (one pass over the CWS table)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(sample_intervals=8):
    """The CPU oracle (oracle/hulk_oracle.c, a literal port of the Go algorithm) timed on this
    box's host cores on a bounded sample of the same workload.  Single thread."""
    from oracle import pyorc
    from hulk_amd import synth
    o = pyorc.Sketcher(K, W, S, 0, 1.0, INTERVAL)          # CWS table generation: not timed
    bases, offsets = synth.reads_numpy(0, sample_intervals * INTERVAL, READ_LEN)
    t0 = time.perf_counter()
    o.add_reads(bases, offsets)
    dt = time.perf_counter() - t0
    n = sample_intervals * INTERVAL
    o.close()
    return {"value": n / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"{n} reads = {sample_intervals} intervals of {INTERVAL} "
                      f"(k={K}, sketchSize={S}), single-threaded C port of the Go path, "
                      f"CWS table generation excluded; {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-pass", action="store_true",
                    help="skip the second timed pass (CWS-scan pruning disabled) that fills value_unpruned")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the all-reduce path even at world size 1 (test aid)")
    args = ap.parse_args()

    # RCCL / the HIP runtime print banners on stdout; the contract is ONE JSON line there.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd.distributed import GpuEngine, ShardedSketcher, slot_shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device(device))

    steps, warmup = args.steps, args.warmup
    total_steps = steps + warmup
    # global interval = world * INTERVAL; this rank bins its contiguous INTERVAL-read slice of each
    reads_per_rank_step = INTERVAL * BATCH
    sb, sc = slot_shard(S, rank, world)

    os.environ["HULK_BATCH"] = str(BATCH)
    # work stream (minimizer kernels) and, for N > 1, a second stream for the collective: the all-reduce
    # of step n and the flush behind it run under the minimizer kernels of step n+1
    stream = torch.cuda.Stream(device=device)
    coll_stream = torch.cuda.Stream(device=device) if use_dist else None
    torch.cuda.set_stream(stream)

    # synthetic reads, resident in HBM.  Interval t of step s = global reads
    # [(s*BATCH+t)*world*INTERVAL, +world*INTERVAL); this rank owns the slice [rank*INTERVAL, +INTERVAL)
    # of it, so an N-rank run sketches the same global stream as a 1-rank run with interval N*100k.
    n_buf = min(total_steps, 24)          # distinct steps kept in HBM (reused cyclically beyond that)
    step_bases, offsets = [], None
    for s_ in range(n_buf):
        parts = []
        for t in range(BATCH):
            first = ((s_ * BATCH + t) * world + rank) * INTERVAL
            b, _ = synth.reads_torch(first, INTERVAL, READ_LEN, device=device)
            parts.append(b[:INTERVAL * READ_LEN])
        pad = torch.zeros(16, dtype=torch.uint8, device=device)
        step_bases.append(torch.cat(parts + [pad]))
    offsets = torch.arange(reads_per_rank_step + 1, dtype=torch.int64, device=device) * READ_LEN
    torch.cuda.synchronize()


    def run_pass(prune):
        """warm-up + the timed K steps on a fresh context; prune=False disables the exact bound test of the
        CWS scan (HULK_NO_PRUNE), so that every interval streams the whole table like the reference does."""
        if prune:
            os.environ.pop("HULK_NO_PRUNE", None)
        else:
            os.environ["HULK_NO_PRUNE"] = "1"
        sk = hulk_amd.GpuSketcher(K, W, S, interval=0, decay_ratio=1.0, device=local_rank,
                                  slot_begin=sb, slot_count=sc, stream=stream.cuda_stream)
        assert sk.batch_size == BATCH
        eng = GpuEngine(sk, device, n_spectra=BATCH)
        sh = ShardedSketcher(eng, S, rank, world if use_dist else 1, dist if use_dist else None)

        def one_step(t):
            b = step_bases[t % n_buf]
            sk.bin_reads_device(b.data_ptr(), offsets.data_ptr(), reads_per_rank_step, READ_LEN, b.numel(),
                                reads_per_spectrum=INTERVAL)
            if use_dist:
                h = eng.histogram_tensor()                 # view of the ring the reads were just binned into
                coll_stream.wait_stream(stream)
                with torch.cuda.stream(coll_stream):
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                sk.flush_batch(BATCH, after_stream=coll_stream.cuda_stream)
            else:
                sk.flush_batch(BATCH)

        for t in range(warmup):
            one_step(t)
        torch.cuda.synchronize()
        sk.set_profiling(True)
        tiles0 = sk.scan_stats()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(warmup, total_steps):
            one_step(t)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        tiles1 = sk.scan_stats()
        n_launch, scan_ms = sk.get_profile("k_cws_scan")
        n_k1, k1_ms = sk.get_profile("k_minimizer_fast")
        sk.set_profiling(False)
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())

        sk.finish()
        counters = sk.counters()
        mins, weights = sh.gather_sketch() if (use_dist and world > 1) else sk.sketch()

        sk.close()
        return dict(elapsed=elapsed, n_launch=n_launch, scan_ms=scan_ms, n_k1=n_k1, k1_ms=k1_ms, counters=counters,
                    mins=mins, weights=weights, tiles0=tiles0, tiles1=tiles1)

    full = run_pass(False) if not args.single_pass else None
    main_pass = run_pass(True)
    elapsed, n_launch, scan_ms, n_k1, k1_ms = (main_pass[k] for k in ("elapsed", "n_launch", "scan_ms", "n_k1", "k1_ms"))
    counters, mins, weights, tiles0, tiles1 = (main_pass[k] for k in ("counters", "mins", "weights", "tiles0", "tiles1"))
    if full is not None and rank == 0:
        assert np.array_equal(full["mins"], mins) and np.array_equal(full["weights"], weights), "pruning changed the sketch"

    if rank == 0:
        total_reads = steps * reads_per_rank_step * world
        value = total_reads / elapsed
        # Per-launch durations measured live with HIP events on the work stream (hulk_set_profiling).
        # Dominant kernel by time = k_minimizer_fast (minimizers + jump hash + spectrum atomics): its
        # algorithmic HBM bytes are the bases (1 B/base) + the read offsets (8 B/read); it is bound by
        # VALU issue (integer hashing, fp64 jump hash), not by HBM — frac is reported against the HBM
        # peak as the contract asks, the VALU-busy fraction from rocprofv3 PMC is in profiles/.
        k1_avg_s = (k1_ms / 1e3) / max(n_k1, 1)
        # algorithmic bytes of k_minimizer_fast: bases + offsets in, (value u64 + slot u8) per distinct
        # minimizer out (the list k_jump_bin consumes)
        per_read_min = counters["n_minimizers"] / float(counters["n_reads"])
        k1_bytes = float(reads_per_rank_step) * (READ_LEN + 8 + 9.0 * per_read_min)
        k1_ach = k1_bytes / k1_avg_s / 1e9 if k1_avg_s > 0 else 0.0
        # The HBM-streaming kernel of the path = k_cws_scan.  Unpruned it makes ONE fp32 pass over this rank's
        # slice of K per launch (4*slots*k^4 bytes) + the BATCH reciprocal vectors; with the exact
        # bound test (no concept drift) it only reads the 8-slot x 256-bin tiles that can still lower a weight,
        # so the bytes it is priced on are the tiles it actually read (hulk_get_scan_stats) + the small tables.
        wtiles = ((K ** 4 + 1023) // 1024) * 4
        full_bytes = 4.0 * sc * (K ** 4) + 4.0 * BATCH * (K ** 4)
        visited = (tiles1[0] - tiles0[0]) / max(n_launch, 1)
        covered = (tiles1[1] - tiles0[1]) / max(n_launch, 1)
        alg_bytes = visited * 8 * 256 * 4.0 + 4.0 * BATCH * (K ** 4) + 4.0 * sc * wtiles + 8.0 * BATCH * wtiles
        avg_s = (scan_ms / 1e3) / max(n_launch, 1)
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        out = {
            "metric": "reads/sec (150bp, k=21, sketch=512)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2: synthetic 150bp reads, k=21, w=9, sketchSize=512, "
                                   f"interval=100k reads per rank, {BATCH} intervals per step, HBM-resident input",
                       "reads_per_step": reads_per_rank_step * world, "total_reads": total_reads,
                       "intervals_per_step": BATCH,
                       "parallelism": f"read-shard x{world}, slot-sharded CWS"},
            "roofline": {"bound": "hbm", "kernel": "k_minimizer_fast", "achieved": k1_ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1_ach / HBM_PEAK_GBS,
                         "traffic": None, "launches": int(n_k1), "avg_launch_us": k1_avg_s * 1e6,
                         "alg_bytes_per_launch": k1_bytes,
                         "note": "dominant by time; VALU-issue bound (SQ_ACTIVE_INST_VALU = 100% of SIMD cycles, "
                                 "profiles/r01_pmc.json), not HBM bound"},
            "roofline_cws_scan": {"bound": "hbm", "kernel": "k_cws_scan", "achieved": achieved,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "traffic": None, "launches": int(n_launch),
                                  "avg_launch_us": avg_s * 1e6, "alg_bytes_per_launch": alg_bytes,
                                  "intervals_per_launch": BATCH, "tiles_read_per_launch": visited,
                                  "tiles_covered_per_launch": covered, "unpruned_bytes_per_launch": full_bytes,
                                  "note": "exact branch-and-bound: a batch whose smallest count-min counter already rules "
                                          "out every slot skips the estimates/scan/resolve, otherwise only tiles whose "
                                          "lower bound can beat a slot's current weight are read (identical sketch; "
                                          "HULK_NO_PRUNE=1 disables both = value_unpruned)"},
            "path_bytes_per_read": READ_LEN + 4.0 * S * (K ** 4) / (INTERVAL * world),
            "sketch_md5": __import__("hashlib").md5(mins.astype("<u8").tobytes()).hexdigest(),
            # same K steps with the exact pruning of the CWS scan switched off: every interval streams the whole
            # table, as the reference's algorithm does (sketch asserted identical)
            "value_unpruned": (total_reads / full["elapsed"]) if full is not None else None,
            "ms_per_step_unpruned": (full["elapsed"] / steps * 1e3) if full is not None else None,
            "n_minimizers_rank0": counters["n_minimizers"],
        }
        # HBM traffic per launch from rocprofv3 PMC (FETCH_SIZE/WRITE_SIZE, separate passes of this same
        # command, gfx950 correction applied — see profiles/r01_pmc.json); cannot be collected live.
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            if world == 1:
                out["roofline"]["traffic"] = pmc["k_minimizer_fast"]["hbm_bytes_per_launch"]
                out["roofline_cws_scan"]["traffic"] = pmc["k_cws_scan"]["hbm_bytes_per_launch"]   # PMC run, same command
        except Exception:
            pass
        # SURVEY.md §8(d) prices the path at L + 4*S*k^4/I bytes per read (one K pass per interval) => a roofline of
        # 1.94e9 reads/s/GPU at C2; this is value / that rate.  The path itself moves far fewer bytes (one K pass
        # per BATCH intervals, and only the tiles that can still change a slot).
        out["path_hbm_frac"] = value * out["path_bytes_per_read"] / 1e9 / (HBM_PEAK_GBS * world)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
