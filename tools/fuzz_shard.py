#!/usr/bin/env python3
"""Randomised differential test of the multi-rank protocol inside libhulkhip (hulk_step_sharded / hulk_step_sliced over the
HOST transport): G ranks = G contexts on the one GPU, each driven by its own thread of this process, connected by an
in-process exchange function (a barrier and shared arrays stand in for the network); the gathered sketch and the count-min
counters must equal the CPU oracle's over the same global stream.  Random k, w, sketch size, interval, batch size T, world
size (1..8: slot shards and ragged last steps of every shape, ranks without any interval), stream length (partial last
interval), decay (concept drift takes the spectra exchange on every step), reads with N, the work lanes of the binning side
(hulk_params.work_lanes).  FUZZ_SHARD_FULL=1 (or the older HULK_SHARD_FULL=1) in the environment forces the
spectra exchange on every step of every case (HULK_FLAG_SHARD_FULL on every context).
usage: fuzz_shard.py [n_cases] [seed]     (run on the GPU box)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before libhulkhip, see hulk_amd/_lib.py)
import hulk_amd
from hulk_amd import _lib
from hulk_amd.distributed import interval_slice, num_steps, slot_shard, step_share
from oracle import pyorc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
FORCE_FULL = bool(os.environ.get("FUZZ_SHARD_FULL") or os.environ.get("HULK_SHARD_FULL"))
MAX_WORLD = int(os.environ.get("FUZZ_MAX_WORLD", "8"))


class Exchange:
    """all-gather / uint32 all-reduce among the threads of this process"""
    def __init__(self, world):
        self.world, self.bar, self.slots = world, threading.Barrier(world), [None] * world

    def make(self, rank):
        def exchange(op, send, recv):
            self.slots[rank] = send.copy()
            self.bar.wait()
            if op == 0:
                recv[:] = np.concatenate(self.slots)
            else:
                acc = np.zeros(len(send) // 4, dtype=np.uint32)
                for s_ in self.slots:
                    acc += s_.view(np.uint32)
                recv[:] = acc.view(np.uint8)
            self.bar.wait()
        return exchange


def reads(rng_, n, L, alph):
    a = np.frombuffer(alph, dtype=np.uint8)
    bases = a[rng_.integers(0, len(a), size=n * L)]
    return bases, np.arange(n + 1, dtype=np.uint64) * np.uint64(L)


bad = 0
health = {"headers_refetched": 0, "void_blocks": 0}                  # hulk_get_comm_health over every rank of every case
t_start = time.time()
budget = float(os.environ.get("FUZZ_SECONDS", 0))                    # stop after this many seconds (the summary counts the cases done)
for case in range(n_cases):
    if budget and time.time() - t_start > budget:
        n_cases = case
        break
    k = int(rng.choice([11, 13, 15, 15, 17, 21]))
    w = int(rng.choice([4, 5, 9, 9, 12]))
    S = int(rng.choice([3, 8, 16, 50]))
    world = int(rng.integers(1, MAX_WORLD + 1))
    T = int(rng.choice([1, 2, 4, 8, 16]))
    I = int(rng.choice([700, 1500, 3000]))
    L = int(rng.choice([60, 100, 150, 260])) if k + 16 * w > 270 else int(rng.choice([60, 100, 150]))
    L = max(L, w + k - 1)
    decay = float(rng.choice([1.0, 1.0, 1.0, 0.02, 0.5]))
    mode = str(rng.choice(["sharded", "sharded", "sharded", "sliced"]))
    alph = [b"ACGT", b"ACGT", b"ACGTN", b"ACGTacgtn"][int(rng.integers(0, 4))]
    if k == 21:                                                      # (the oracle's flush is S * k^4 evaluations per interval)
        S, T = min(S, 8), min(T, 4)
    n_int = int(rng.integers(1, min(3 * world * T + 3, 70)))         # intervals of the global stream (the last may be partial)
    total = n_int * I - int(rng.integers(0, I)) if rng.random() < 0.6 else n_int * I
    total = max(total, 1)
    if mode == "sliced":
        total = n_int * I                                            # (whole intervals: every rank slices every interval)
    bases, offsets = reads(rng, total, L, alph)
    lanes = int(rng.choice([1, 2, 2]))                                # work lanes of the binning side
    ex = Exchange(world)
    out, errs = [None] * world, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            sb, sc = slot_shard(S, rank, world)
            sk = hulk_amd.GpuSketcher(k, w, S, interval=(0 if mode == "sliced" else I), decay_ratio=decay, device=0,
                                      slot_begin=sb, slot_count=sc, batch=T, work_lanes=lanes,
                                      flags=_lib.HULK_FLAG_SHARD_FULL if FORCE_FULL else 0)
            sk.comm_init_host(rank, world, ex.make(rank))
            if mode != "sliced":
                for s_ in range(num_steps(total, T, I, world)):
                    first, n, si = step_share(s_, T, I, rank, world, total)
                    lo, hi = int(offsets[first]) if n else 0, int(offsets[first + n]) if n else 0
                    sk.step_sharded_host(bases[lo:hi], offsets[first:first + n + 1] - offsets[first], si) if n else \
                        sk.step_sharded_host(np.zeros(0, np.uint8), np.zeros(1, np.uint64), si)
            else:
                tb = torch.from_numpy(bases).cuda()
                for b0 in range(0, n_int, T):
                    nt = min(T, n_int - b0)
                    parts, cnt0 = [], None
                    for t in range(nt):
                        first, cnt = interval_slice("strong", b0 + t, I, rank, world)
                        parts.append(tb[first * L:(first + cnt) * L]); cnt0 = cnt
                    per = interval_slice("strong", 0, I, rank, world)[1]
                    buf = torch.cat(parts + [torch.zeros(16, dtype=torch.uint8, device="cuda")])
                    off = torch.arange(per * nt + 1, dtype=torch.int64, device="cuda") * L
                    torch.cuda.synchronize()
                    sk.step_sliced(buf.data_ptr(), off.data_ptr(), per * nt, L, buf.numel(), per, nt)
                    sk.synchronize()
            err = None
            try:
                sk.finish()
            except hulk_amd.HulkError as e:
                err = str(e)
            m, wt = sk.gather_sketch()
            out[rank] = (m, wt, sk.cms(), err, sk.comm_stats())
            sk.close()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            try:
                ex.bar.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for x in out:
        if x is not None:
            for key in health:
                health[key] += x[4].get(key, 0)
    o = pyorc.Sketcher(k, w, S, 0, decay, I)
    oerr = None
    try:
        o.add_reads(bases, offsets)
        o.finish()
    except pyorc.OracleError as e:
        oerr = str(e)
    desc = f"case {case}: k={k} w={w} S={S} world={world} T={T} I={I} L={L} decay={decay} mode={mode} total={total} alph={alph!r}"
    if errs:
        bad += 1
        print("MISMATCH (exception)", desc, errs)
    elif oerr or any(x[3] for x in out):
        if not (oerr and all(x[3] for x in out)):
            bad += 1
            print("MISMATCH (error)", desc, oerr, [x[3] for x in out])
    else:
        om, ow = o.sketch()
        ocms = o.cms()
        for r_, (m, wt, cms, _, st) in enumerate(out):
            ok = np.array_equal(m, om) and np.allclose(wt, ow, rtol=1e-9 if decay == 1.0 else 1e-7, atol=0) and \
                (np.array_equal(cms, ocms) if decay == 1.0 or decay == 0.0 else np.allclose(cms, ocms, rtol=1e-9, atol=1e-300))
            if not ok:
                bad += 1
                print("MISMATCH", desc, "rank", r_, st, int((m != om).sum()), "mins differ")
                break
    o.close()
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t_start:.1f} s (seed {seed}); exchange headers fetched again: "
      f"{health['headers_refetched']}, void blocks: {health['void_blocks']}")
sys.exit(1 if bad else 0)
