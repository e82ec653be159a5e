"""Host side of the multi-GPU path on CPU (hulk_amd/distributed.py), world size 2 over gloo.

The exchange itself lives in libhulkhip.so (hulk_step_sharded; it needs a GPU: tests/test_gpu_two_rank.py runs two product
ranks on one MI355X over the same gloo transport).  What can be checked without one:
  * which reads / slots a rank takes (step_share, interval_slice, slot_shard) tile the global stream;
  * the exchange function the library calls on a host transport (gloo_exchange), both operations, between two processes;
  * the two claims hulk_step_sharded's delta exchange rests on, against the CPU oracle (tests are the only place where the
    oracle may stand in for the product): (1) once every rank's whole-step bound has said "no element can lower a
    weight" one step earlier, flushing the step's intervals does not change the sketch; (2) the count-min counters after
    the step are the counters before it plus the per-interval increments the ranks exchange.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hulk_amd import synth
from hulk_amd.distributed import gloo_exchange, interval_slice, num_steps, read_shard, slot_shard, step_share

K, W, S, I, T, L = 15, 9, 10, 2000, 2, 100
WORLD = 2
TOTAL = 5 * WORLD * T * I + I + 700          # five whole steps + a ragged one: a whole interval and a partial one on rank 0, none on rank 1
HDR = 32


class ModelRank:
    """Host-language model of hulk_step_sharded (include/hulk_hip.h), arithmetic by the CPU oracle.  The oracle cannot
    advance its count-min counters without AddElement, so a delta step ALSO fetches the spectra and flushes them — and
    asserts that this changed nothing but the counters, by exactly the exchanged increments."""

    def __init__(self, rank, world, exchange):
        from oracle import pyorc
        self.pyorc, self.rank, self.world, self.x = pyorc, rank, world, exchange
        self.o = pyorc.Sketcher(K, W, S, 0, 1.0, 0)
        self.binner = pyorc.Sketcher(K, W, 1, 0, 1.0, 0)
        self.B = K ** 4
        r, c, b = self.o.cws()
        self.kmin = (c * np.exp(b - r)).min(axis=1)                   # min_row(K), K = c * exp(b - r)
        self.lo, self.n = slot_shard(S, rank, world)
        d, g = pyorc.cms_geometry()
        self.depth, self.width = d, g
        bins = np.arange(self.B, dtype=np.uint64)
        self.pos = np.array([[pyorc.jump(int(x + dd * x), g) for x in bins] for dd in range(d)])   # countmin.go:122-125
        self.step_no, self.prev_flags = 0, None
        self.steps_delta = self.steps_full = 0

    def verdict(self):
        """k_flush_decide: can an element of a step that starts now still lower one of this rank's weights?"""
        m = self.o.cms().min()
        if m == 0:
            return 1
        _, w = self.o.sketch()
        for s_ in range(self.lo, self.lo + self.n):
            thr = w[s_] + 1e-5 * abs(w[s_]) + 1e-37
            bound = self.kmin[s_] / m if self.kmin[s_] < 0 else 0.0
            if bound <= thr:
                return 1
        return 0

    def gather(self, arr):
        send = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        recv = np.zeros(send.size * self.world, dtype=np.uint8)
        self.x(0, send, recv)
        return recv.view(arr.dtype).reshape((self.world,) + arr.shape)

    def step(self, bases, offsets, n, step_intervals):
        own = min(T, max(0, step_intervals - self.rank * T))
        assert (own == 0) == (n == 0) and n <= own * I
        spectra = np.zeros((T, self.B), dtype=np.uint32)
        for t in range(own):
            a, b = t * I, min((t + 1) * I, n)
            self.binner.add_reads(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a])
            spectra[t] = self.binner.histogram().astype(np.uint32)
            self.binner.wipe()
        full = self.step_no == 0 or any(self.prev_flags)
        hdr = np.zeros(HDR, dtype=np.uint32)
        hdr[1] = self.verdict()
        counts = [min(T, max(0, step_intervals - r * T)) for r in range(self.world)]
        if not full:
            delta = np.zeros((T, self.depth, self.width), dtype=np.uint32)
            for t in range(own):
                hdr[2 + t] = np.count_nonzero(spectra[t])
                for d in range(self.depth):
                    np.add.at(delta[t, d], self.pos[d], spectra[t])
            hdrs, deltas = self.gather(hdr), self.gather(delta)
            want = self.o.cms().copy()
            for r in range(self.world):
                for t in range(counts[r]):
                    used = int(hdrs[r][2 + t])
                    if used and not used / self.B < 0.01:
                        want += deltas[r][t]
            before = self.o.sketch()
            self.steps_delta += 1
        else:
            hdrs = self.gather(hdr)
            self.steps_full += 1
        allspec = self.gather(spectra)                      # (delta step: the checker's copy, not part of the protocol)
        for r in range(self.world):
            for t in range(counts[r]):
                self.o.add_histogram(allspec[r][t])
                self.o.flush()
        if not full:
            after = self.o.sketch()
            assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1]), "a delta step changed the sketch"
            assert np.array_equal(self.o.cms(), want), "count-min counters != counters + exchanged increments"
        self.prev_flags = [int(h[1]) for h in hdrs]
        self.step_no += 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = gloo_exchange(dist)
    # the exchange function by itself, as the library calls it: uint8 views of its staging
    send = np.arange(12, dtype=np.uint32) + 100 * rank
    recv = np.zeros(12 * world, dtype=np.uint32)
    x(0, send.view(np.uint8), recv.view(np.uint8))
    assert np.array_equal(recv, np.concatenate([np.arange(12, dtype=np.uint32) + 100 * r for r in range(world)]))
    big = np.full(5, 0xFFFFFFF0 + rank, dtype=np.uint32)          # the sum wraps like uint32 addition
    out = np.zeros(5, dtype=np.uint32)
    x(1, big.view(np.uint8), out.view(np.uint8))
    assert np.array_equal(out, np.full(5, (sum(0xFFFFFFF0 + r for r in range(world))) & 0xFFFFFFFF, dtype=np.uint32))
    m = ModelRank(rank, world, x)
    for s_ in range(num_steps(TOTAL, T, I, world)):
        first, n, step_intervals = step_share(s_, T, I, rank, world, TOTAL)
        bases, offsets = synth.reads_numpy(first, n, L) if n else (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
        m.step(bases, offsets, n, step_intervals)
    mins, weights = m.o.sketch()
    q.put((rank, mins, weights, m.steps_delta, m.steps_full))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_shard_helpers():
    for world in (1, 2, 3, 8):
        cov = []
        for r in range(world):
            b, c = slot_shard(512, r, world); cov += list(range(b, b + c))
        assert cov == list(range(512))
        cov = []
        for r in range(world):
            lo, hi = read_shard(100000, r, world); cov += [(lo, hi)]
        assert cov[0][0] == 0 and cov[-1][1] == 100000 and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))


def test_interval_slices_partition_the_global_stream():
    """strong: the ranks' slices tile interval t = [t*I, (t+1)*I) of the global stream (pipeline/sketch.go:211-215 with
    the reference's interval); weak: they tile [t*G*I, (t+1)*G*I) — a G-times longer interval."""
    for world in (1, 2, 3, 8):
        for t in (0, 1, 7):
            for mode, span in (("strong", 100000), ("weak", 100000 * world)):
                sl = [interval_slice(mode, t, 100000, r, world) for r in range(world)]
                assert sl[0][0] == t * span and sum(c for _, c in sl) == span
                assert all(a[0] + a[1] == b[0] for a, b in zip(sl, sl[1:]))
    with pytest.raises(ValueError):
        interval_slice("sideways", 0, 10, 0, 1)


def test_batch_shares_partition_a_batch_by_whole_intervals():
    """batch_share: the ranks' chunks tile the `batch` intervals of a step in rank order, each a whole number of
    intervals that starts at the spectrum it reports; a batch that does not divide among the ranks is refused."""
    from hulk_amd.distributed import batch_share
    I, batch = 100000, 16
    for world in (1, 2, 4, 8, 16):
        for step in (0, 3):
            sh = [batch_share(step, batch, I, r, world) for r in range(world)]
            assert sh[0][0] == step * batch * I and sum(n for _, n, _ in sh) == batch * I
            assert all(a[0] + a[1] == b[0] for a, b in zip(sh, sh[1:]))
            assert all(n % I == 0 and first == (step * batch + spec) * I for first, n, spec in sh)
    with pytest.raises(ValueError):
        batch_share(0, 16, I, 0, 3)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher spawns the N ranks itself; it must never quietly time one GPU.
    Here (no GPU) it has to exit non-zero with a message and print no JSON line."""
    import subprocess, sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr and "{" not in p.stdout


def test_step_shares_tile_the_stream_by_whole_intervals():
    """step_share: the ranks' chunks of a step tile its G*T intervals in rank order (whole intervals, T per rank); with an
    end of stream the last step is ragged and every rank reports the same step_intervals."""
    I_, T_ = 100000, 16
    for world in (1, 2, 4, 8):
        for step in (0, 3):
            sh = [step_share(step, T_, I_, r, world) for r in range(world)]
            assert sh[0][0] == step * world * T_ * I_ and all(n == T_ * I_ and si == world * T_ for _, n, si in sh)
            assert all(a[0] + a[1] == b[0] for a, b in zip(sh, sh[1:]))
    total = 400_000_000                                    # BASELINE C4 on 8 ranks: 31 whole steps + 32 intervals
    assert num_steps(total, T_, I_, 8) == 32
    last = [step_share(31, T_, I_, r, 8, total) for r in range(8)]
    assert [n for _, n, _ in last] == [1_600_000, 1_600_000, 0, 0, 0, 0, 0, 0] and all(si == 32 for _, _, si in last)
    covered = sum(step_share(s_, T_, I_, r, 8, total)[1] for s_ in range(32) for r in range(8))
    assert covered == total
    # a partial last interval (the reference's EOF flush, pipeline/sketch.go:219-221) stays with the rank that holds it
    first, n, si = step_share(0, 4, 1000, 1, 2, 4000 + 2500)
    assert (first, n, si) == (4000, 2500, 7)
    assert step_share(5, 4, 1000, 0, 2, 6500) == (40000, 0, 0)


@pytest.mark.timeout(600)
def test_two_ranks_protocol_model_against_the_oracle():
    from oracle import pyorc
    ref = pyorc.Sketcher(K, W, S, 0, 1.0, I)
    bases, offsets = synth.reads_numpy(0, TOTAL, L)
    ref.add_reads(bases, offsets); ref.finish()
    rm, rw = ref.sketch()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs: p.start()
    outs = [q.get(timeout=500) for _ in range(WORLD)]
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, mins, weights, n_delta, n_full in outs:
        assert np.array_equal(mins, rm) and np.array_equal(weights, rw), f"rank {rank}"
        assert n_full >= 1 and n_delta >= 1, (n_delta, n_full)          # both exchanges were taken
