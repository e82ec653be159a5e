"""Two ranks of the PRODUCT multi-GPU path (GpuSketcher + GpuEngine + ShardedSketcher, the protocol of
bench.py) on ONE MI355X: both processes use cuda:0 and exchange over gloo (RCCL refuses two ranks on
one device; the collective is the only thing swapped).  The gathered sketch must equal a single-rank
GPU run over the same global read stream, and the oracle:
  * "strong" (bench.py's default): each rank bins its half of every interval of I reads — the single-rank
    run has the SAME interval I (the reference's rule, pipeline/sketch.go:211-215);
  * "weak": each rank bins I reads per interval — the single-rank run has interval 2 x I;
  * "strong-interval": the strong rule split the other way (distributed.batch_share): each rank bins WHOLE intervals of a
    batch through hulk_bin_reads_device_at(first_spectrum), the all-reduce over the ring is a gather.
The stream ends in a ragged tail (TAIL < BATCH intervals binned, then finish() without a flush_batch): the final flush
must take every spectrum of the tail batch.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, W, S, I, BATCH, STEPS, TAIL, L = 15, 9, 64, 3000, 4, 3, 2, 150


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q, overlap, scaling):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["HULK_BATCH"] = str(BATCH)
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd.distributed import GpuEngine, ShardedSketcher, batch_share, interval_slice, slot_shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    coll = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sb, sc = slot_shard(S, rank, world)
    sk = hulk_amd.GpuSketcher(K, W, S, interval=0, decay_ratio=1.0, device=0, slot_begin=sb, slot_count=sc,
                              stream=stream.cuda_stream)
    assert sk.batch_size == BATCH
    eng = GpuEngine(sk, "cuda:0", n_spectra=BATCH)
    sh = ShardedSketcher(eng, S, rank, world, dist)
    by_interval = scaling == "strong-interval"              # whole intervals of a batch per rank (distributed.batch_share)
    per = I if by_interval else interval_slice(scaling, 0, I, rank, world)[1]     # reads of an interval this rank bins
    offsets = torch.arange(per * BATCH + 1, dtype=torch.int64, device="cuda:0") * L
    keep = []
    for s_ in range(STEPS + 1):
        nt = BATCH if s_ < STEPS else TAIL                  # the last batch is a ragged tail
        parts = []
        first_spectrum = 0
        if by_interval:
            first, cnt, first_spectrum = batch_share(0, nt, I, rank, world)     # this rank's intervals of a batch of nt ...
            first += s_ * BATCH * I                                             # ... that starts at interval s_ * BATCH
            b, _ = synth.reads_torch(first, cnt, L, device="cuda:0")
            parts.append(b[:cnt * L])
            n_mine = cnt
        else:
            for t in range(nt):
                first, cnt = interval_slice(scaling, s_ * BATCH + t, I, rank, world)
                b, _ = synth.reads_torch(first, cnt, L, device="cuda:0")
                parts.append(b[:cnt * L])
            n_mine = per * nt
        bases = torch.cat(parts + [torch.zeros(16, dtype=torch.uint8, device="cuda:0")])
        keep.append(bases)
        sk.bin_reads_device(bases.data_ptr(), offsets.data_ptr(), n_mine, L, bases.numel(), reads_per_spectrum=per,
                            first_spectrum=first_spectrum)
        h = eng.histogram_tensor()
        if s_ == STEPS:                                     # tail: all-reduce the whole view, then finish() flushes it
            torch.cuda.synchronize()
            hc = h.cpu()
            dist.all_reduce(hc, op=dist.ReduceOp.SUM)
            h.copy_(hc)
            torch.cuda.synchronize()
            break
        if overlap:                                     # bench.py's shape: collective + flush on a second stream
            coll.wait_stream(stream)
            with torch.cuda.stream(coll):
                hc = h.cpu()                            # gloo: host-staged all-reduce of the int32 spectra
                dist.all_reduce(hc, op=dist.ReduceOp.SUM)
                h.copy_(hc)
            sk.flush_batch(BATCH, after_stream=coll.cuda_stream)
        else:
            hc = h.cpu()
            dist.all_reduce(hc, op=dist.ReduceOp.SUM)
            h.copy_(hc)
            sk.flush_batch(BATCH)
    sk.finish()
    eng.collective_device = None                        # gloo gather on host tensors
    mins, weights = sh.gather_sketch()
    if rank == 0:
        q.put((mins, weights, sk.counters()))
    sk.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,scaling", [(False, "weak"), (True, "weak"), (False, "strong"), (True, "strong"),
                                             (False, "strong-interval"), (True, "strong-interval")])
def test_two_ranks_one_gpu_match_single_rank(overlap, scaling):
    import torch
    import torch.multiprocessing as mp
    import hulk_amd
    from hulk_amd import synth
    from oracle import pyorc
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, overlap, scaling)) for r in range(world)]
    for p in procs:
        p.start()
    mins, weights, _ = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single rank, same global stream: interval = I (strong: the reference's rule) or world * I (weak)
    gi = world * I if scaling == "weak" else I
    total = (STEPS * BATCH + TAIL) * gi
    bases, offsets = synth.reads_numpy(0, total, L)
    g = hulk_amd.GpuSketcher(K, W, S, interval=gi)
    g.add_reads(bases, offsets)
    g.finish()
    m1, w1 = g.sketch()
    g.close()
    assert np.array_equal(mins, m1)
    assert np.array_equal(weights, w1)                  # same kernels, same order: bit-identical
    o = pyorc.Sketcher(K, W, S, 0, 1.0, gi)
    o.add_reads(bases, offsets)
    o.finish()
    mo, wo = o.sketch()
    o.close()
    assert np.array_equal(mins, mo)
    assert np.allclose(weights, wo, rtol=1e-12, atol=0)
