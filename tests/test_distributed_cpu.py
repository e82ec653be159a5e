"""World-size-2 test of the multi-GPU host logic (hulk_amd/distributed.py) on CPU over gloo.
The GPU engine is replaced by a test double built on the CPU oracle — tests are the only place
where that is allowed; the product engine (GpuEngine over libhulkhip) has no such fallback."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hulk_amd import synth
from hulk_amd.distributed import ShardedSketcher, interval_slice, read_shard, slot_shard

K, W, S, I, NI, L = 9, 4, 10, 600, 3, 80


class OracleEngine:
    """Test double: same duck type as GpuEngine."""
    def __init__(self, rank, world):
        from oracle import pyorc
        self.pyorc = pyorc
        self.o = pyorc.Sketcher(K, W, S)
        self.hist = torch.zeros(K ** 4, dtype=torch.int32)
        self.lo, self.n = slot_shard(S, rank, world)

    def bin_reads(self, bases, offsets):
        for i in range(len(offsets) - 1):
            seq = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
            for x in self.pyorc.minimizers(seq, K, W):
                self.hist[self.pyorc.jump(int(x), K ** 4)] += 1

    def histogram_tensor(self): return self.hist

    def flush(self):
        self.o.add_histogram(self.hist.numpy().astype(np.uint32))
        self.o.flush()
        self.hist.zero_()

    def finish(self):
        self.flush()

    def sketch(self):
        m, w = self.o.sketch()
        mm = np.zeros(S, dtype=np.uint64); ww = np.full(S, np.finfo(np.float64).max)
        mm[self.lo:self.lo + self.n] = m[self.lo:self.lo + self.n]      # only the owned slots
        ww[self.lo:self.lo + self.n] = w[self.lo:self.lo + self.n]
        return mm, ww


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = OracleEngine(rank, world)
    sh = ShardedSketcher(eng, S, rank, world, dist)
    for t in range(NI):
        first, cnt = interval_slice("strong", t, I, rank, world)     # a slice of the GLOBAL interval
        bases, offsets = synth.reads_numpy(first, cnt, L)
        eng.bin_reads(bases, offsets)
        sh.end_interval()
    sh.finish()
    mins, weights = sh.gather_sketch()
    q.put((rank, mins, weights))
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


@pytest.mark.timeout(300)
def test_two_ranks_equal_single_process():
    from oracle import pyorc
    ref = pyorc.Sketcher(K, W, S, 0, 1.0, I)
    bases, offsets = synth.reads_numpy(0, NI * I, L)
    ref.add_reads(bases, offsets); ref.finish()
    rm, rw = ref.sketch()

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs: p.join(60)
    for rank, mins, weights in outs:
        assert np.array_equal(mins, rm), f"rank {rank}"
        assert np.array_equal(weights, rw), f"rank {rank}"


NB = 2          # batches of BatchOracleEngine.T intervals in the whole-interval test


class BatchOracleEngine(OracleEngine):
    """Test double of a batched engine (hulk_bin_reads_device_at): T spectra per exchange, a rank fills the ones it owns."""
    T = 2

    def __init__(self, rank, world):
        super().__init__(rank, world)
        self.hist = torch.zeros(self.T * K ** 4, dtype=torch.int32)

    def bin_reads_at(self, bases, offsets, reads_per_spectrum, first_spectrum):
        B = K ** 4
        for i in range(len(offsets) - 1):
            seq = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
            t = first_spectrum + i // reads_per_spectrum
            for x in self.pyorc.minimizers(seq, K, W):
                self.hist[t * B + self.pyorc.jump(int(x), B)] += 1

    def flush(self):
        B = K ** 4
        for t in range(self.T):                         # the spectra of the batch, in interval order
            self.o.add_histogram(self.hist[t * B:(t + 1) * B].numpy().astype(np.uint32))
            self.o.flush()
        self.hist.zero_()

    def finish(self):
        pass                                            # (every batch of this test is complete)


def _worker_whole_intervals(rank, world, port, q):
    from hulk_amd.distributed import batch_share
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = BatchOracleEngine(rank, world)
    sh = ShardedSketcher(eng, S, rank, world, dist)
    for step in range(NB):
        first, cnt, first_spec = batch_share(step, eng.T, I, rank, world)    # whole intervals of the batch
        bases, offsets = synth.reads_numpy(first, cnt, L)
        eng.bin_reads_at(bases, offsets, I, first_spec)
        sh.end_interval()                               # ONE all-reduce over the T spectra (a gather here), then the flush
    sh.finish()
    mins, weights = sh.gather_sketch()
    q.put((rank, mins, weights))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_whole_intervals_equal_single_process():
    """The strong rule shared by whole intervals (distributed.batch_share): rank g fills the spectra of ITS intervals of a
    batch, one all-reduce over the batch's spectra gathers them, the flush takes them in interval order — the sketch of one
    process with the same interval."""
    from oracle import pyorc
    ref = pyorc.Sketcher(K, W, S, 0, 1.0, I)
    bases, offsets = synth.reads_numpy(0, NB * BatchOracleEngine.T * I, L)
    ref.add_reads(bases, offsets); ref.finish()
    rm, rw = ref.sketch()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_whole_intervals, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs: p.join(60)
    for rank, mins, weights in outs:
        assert np.array_equal(mins, rm), f"rank {rank}"
        assert np.array_equal(weights, rw), f"rank {rank}"
