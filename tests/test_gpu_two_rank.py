"""Two ranks of the PRODUCT multi-GPU path on ONE MI355X: both processes use cuda:0, every step goes through
hulk_step_sharded / hulk_step_sliced (the exchange inside libhulkhip.so) and the ranks are connected by the library's HOST
transport over gloo (hulk_comm_init_host + distributed.gloo_exchange) — RCCL refuses two ranks on one device; the transport
is the only thing swapped, the protocol (which exchange a step takes, the kernels, the order) is the one RCCL ranks run.
The gathered sketch must equal a single-rank GPU run over the same global read stream, and the oracle:
  * "sharded":       whole intervals per rank (distributed.step_share); the first step exchanges spectra, later ones the
                     count-min increments — both exchanges must have been taken;
  * "sharded-full":  the same with HULK_SHARD_FULL=1 (every step exchanges spectra): the delta exchange changes nothing;
  * "sliced-strong": SURVEY.md §8(e) to the letter — each rank bins its half of every interval, one all-reduce per step;
  * "sliced-weak":   each rank bins I reads per interval — the single-rank run has interval 2 x I.
The stream ends in a ragged step (rank 0: a whole and a partial interval, rank 1: nothing) / a ragged tail batch.
RCCL itself runs at world size 1 (test_rccl_world_one_is_the_single_rank_sketch; tests/test_gpu_cpp_host.py from C++).
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, W, S, I, BATCH, L = 15, 9, 64, 3000, 4, 150
WORLD = 2
TOTAL = 4 * WORLD * BATCH * I + I + 1100      # four whole steps + a ragged one
STEPS, TAIL = 3, 2                            # sliced modes: whole batches + a tail batch of TAIL intervals


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q, overlap, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd.distributed import gloo_exchange, interval_slice, num_steps, slot_shard, step_share
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sb, sc = slot_shard(S, rank, world)
    sharded = mode.startswith("sharded")
    from hulk_amd import _lib
    flags = (0 if overlap else _lib.HULK_FLAG_NO_OVERLAP) | (_lib.HULK_FLAG_SHARD_FULL if mode == "sharded-full" else 0)
    sk = hulk_amd.GpuSketcher(K, W, S, interval=I if sharded else 0, decay_ratio=1.0, device=0, slot_begin=sb, slot_count=sc,
                              batch=BATCH, flags=flags)
    assert sk.batch_size == BATCH
    sk.comm_init_host(rank, world, gloo_exchange(dist))
    keep = []
    if sharded:
        for s_ in range(num_steps(TOTAL, BATCH, I, world)):
            first, n, step_intervals = step_share(s_, BATCH, I, rank, world, TOTAL)
            b, off = synth.reads_torch(first, max(n, 1), L, device="cuda:0")
            keep.append((b, off))
            torch.cuda.synchronize()
            sk.step_sharded(b.data_ptr(), off.data_ptr(), n, L, b.numel(), step_intervals)
    else:
        scaling = mode.split("-")[1]
        per = interval_slice(scaling, 0, I, rank, world)[1]
        offsets = torch.arange(per * BATCH + 1, dtype=torch.int64, device="cuda:0") * L
        for s_ in range(STEPS + 1):
            nt = BATCH if s_ < STEPS else TAIL
            parts = []
            for t in range(nt):
                first, cnt = interval_slice(scaling, s_ * BATCH + t, I, rank, world)
                b, _ = synth.reads_torch(first, cnt, L, device="cuda:0")
                parts.append(b[:cnt * L])
            bases = torch.cat(parts + [torch.zeros(16, dtype=torch.uint8, device="cuda:0")])
            keep.append(bases)
            torch.cuda.synchronize()
            sk.step_sliced(bases.data_ptr(), offsets.data_ptr(), per * nt, L, bases.numel(), per, nt)
    sk.finish()
    mins, weights = sk.gather_sketch()
    stats = sk.comm_stats()
    if rank == 0:
        q.put((mins, weights, sk.cms(), stats))
    sk.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,mode", [(True, "sharded"), (False, "sharded"), (True, "sharded-full"),
                                          (True, "sliced-strong"), (False, "sliced-strong"), (True, "sliced-weak")])
def test_two_ranks_one_gpu_match_single_rank(overlap, mode):
    import torch
    import torch.multiprocessing as mp
    import hulk_amd
    from hulk_amd import synth
    from oracle import pyorc
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = WORLD
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, overlap, mode)) for r in range(world)]
    for p in procs:
        p.start()
    mins, weights, cms, stats = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if mode == "sharded":
        assert stats["steps_full"] >= 1 and stats["steps_delta"] >= 1, stats      # both exchanges were taken
    elif mode == "sharded-full":
        assert stats["steps_delta"] == 0
    # single rank, same global stream: interval = I (the reference's rule) or world * I (weak)
    gi = world * I if mode == "sliced-weak" else I
    total = TOTAL if mode.startswith("sharded") else (STEPS * BATCH + TAIL) * gi
    bases, offsets = synth.reads_numpy(0, total, L)
    g = hulk_amd.GpuSketcher(K, W, S, interval=gi)
    g.add_reads(bases, offsets)
    g.finish()
    m1, w1 = g.sketch()
    c1 = g.cms()
    g.close()
    assert np.array_equal(mins, m1)
    assert np.array_equal(weights, w1)                  # same kernels, same order: bit-identical
    assert np.array_equal(cms, c1)                      # count-min counters: integer sums, whichever exchange moved them
    o = pyorc.Sketcher(K, W, S, 0, 1.0, gi)
    o.add_reads(bases, offsets)
    o.finish()
    mo, wo = o.sketch()
    assert np.array_equal(o.cms(), cms)
    o.close()
    assert np.array_equal(mins, mo)
    assert np.allclose(weights, wo, rtol=1e-12, atol=0)


def test_rccl_world_one_is_the_single_rank_sketch():
    """hulk_comm_unique_id + hulk_comm_init bind librccl.so.1 and build a communicator (world size 1 on this box); the
    sharded step over it — spectra exchange first, increments later — gives the sketch of hulk_add_reads."""
    import torch
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd.distributed import num_steps, step_share
    total = 5 * BATCH * I + 700
    sk = hulk_amd.GpuSketcher(K, W, S, interval=I, batch=BATCH)
    sk.comm_init(hulk_amd.GpuSketcher.comm_unique_id(), 0, 1)
    keep = []
    for s_ in range(num_steps(total, BATCH, I, 1)):
        first, n, si = step_share(s_, BATCH, I, 0, 1, total)
        b, off = synth.reads_torch(first, n, L, device="cuda:0")
        keep.append((b, off))
        torch.cuda.synchronize()
        sk.step_sharded(b.data_ptr(), off.data_ptr(), n, L, b.numel(), si)
    sk.finish()
    mins, weights = sk.gather_sketch()
    stats = sk.comm_stats()
    cms, cnt = sk.cms(), sk.counters()
    sk.close()
    assert stats["steps_full"] >= 1 and stats["steps_delta"] >= 1, stats
    bases, offsets = synth.reads_numpy(0, total, L)
    g = hulk_amd.GpuSketcher(K, W, S, interval=I)
    g.add_reads(bases, offsets)
    g.finish()
    m1, w1 = g.sketch()
    assert np.array_equal(mins, m1) and np.array_equal(weights, w1) and np.array_equal(cms, g.cms())
    assert cnt == g.counters()
    g.close()


def test_sharded_step_reports_too_few_used_bins():
    """The 1 % rule (kmerspectrum.go:84-96, "not used yet") must fire on the delta exchange too: an interval of identical
    reads uses a handful of bins."""
    import torch
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd._lib import HulkError
    sk = hulk_amd.GpuSketcher(K, W, S, interval=I, batch=BATCH)
    sk.comm_init_loopback(0, 1)
    keep = []
    for s_ in range(4):
        b, off = synth.reads_torch(s_ * BATCH * I, BATCH * I, L, device="cuda:0")
        if s_ == 3:
            b = b.clone()
            b[:BATCH * I * L] = b[:L].repeat(BATCH * I)            # every read of the step = the first one
        keep.append((b, off))
        torch.cuda.synchronize()
        sk.step_sharded(b.data_ptr(), off.data_ptr(), BATCH * I, L, b.numel(), BATCH)
    assert sk.comm_stats()["steps_delta"] >= 1
    with pytest.raises(HulkError) as e:
        sk.finish()
    assert "not used yet" in str(e.value)
    sk.close()


def test_sharded_step_from_host_slices():
    """hulk_step_sharded_host (a Go host's byte slices: validated like hulk_add_reads, staged through pinned memory) gives the
    sketch of the device-pointer call; a read shorter than w + k - 1 is refused with the reference's text before anything runs."""
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd._lib import HulkError
    from hulk_amd.distributed import num_steps, step_share
    total = 3 * BATCH * I + 2 * I + 500
    sk = hulk_amd.GpuSketcher(K, W, S, interval=I, batch=BATCH)
    sk.comm_init_loopback(0, 1)
    for s_ in range(num_steps(total, BATCH, I, 1)):
        first, n, si = step_share(s_, BATCH, I, 0, 1, total)
        bases, offsets = synth.reads_numpy(first, n, L)
        sk.step_sharded_host(bases, offsets, si)
    sk.finish()
    mins, weights = sk.gather_sketch()
    cms = sk.cms()
    sk.close()
    bases, offsets = synth.reads_numpy(0, total, L)
    g = hulk_amd.GpuSketcher(K, W, S, interval=I)
    g.add_reads(bases, offsets)
    g.finish()
    m1, w1 = g.sketch()
    assert np.array_equal(mins, m1) and np.array_equal(weights, w1) and np.array_equal(cms, g.cms())
    g.close()
    bad = hulk_amd.GpuSketcher(K, W, S, interval=I)
    bad.comm_init_loopback(0, 1)
    with pytest.raises(HulkError, match="sequence length must be >= w \\+ k - 1"):
        bad.step_sharded_host(np.frombuffer(b"ACGTACGTAC", dtype=np.uint8), np.array([0, 10], dtype=np.uint64), 1)
    bad.close()
