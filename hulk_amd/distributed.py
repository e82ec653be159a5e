"""Multi-GPU sharding of the sketch path: one process per GPU, the exchange inside libhulkhip.so
(include/hulk_hip.h: hulk_comm_init + hulk_step_sharded / hulk_step_sliced + hulk_gather_sketch; RCCL over xGMI, or a
transport of the host).  This module is what a host needs besides those calls: which reads a rank takes, which sketch
slots it owns, and — for hosts whose ranks are connected by torch.distributed rather than RCCL (the test suite: two ranks on
one GPU over gloo) — the exchange function hulk_comm_init_host wants.

The interval rule is the reference's throughout (pipeline/sketch.go:211-215: a flush every `interval` reads of the GLOBAL
stream), so an N-rank run computes the sketch of ONE rank over the same stream:
  * whole intervals per rank (hulk_step_sharded, `step_share`): step s covers the global intervals [s*G*T, (s+1)*G*T),
    rank g bins [s*G*T + g*T, s*G*T + (g+1)*T) — one contiguous chunk of T*interval reads.  Count-min is replicated, the CWS
    update is slot-sharded (`slot_shard`); per step ONE all-gather: of the k-mer spectra while an element can still change
    a weight somewhere, of the count-min increments (7 x 2000 integers per interval) once none can;
  * a slice of every interval per rank (hulk_step_sliced, `interval_slice`; SURVEY.md §8e to the letter): ONE all-reduce
    (uint32 sum) of the T spectra of a step.
"""
import numpy as np


def slot_shard(sketch_size: int, rank: int, world: int):
    """Contiguous slot range owned by `rank`: [begin, begin+count)."""
    begin = (sketch_size * rank) // world
    end = (sketch_size * (rank + 1)) // world
    return begin, end - begin


def read_shard(interval_reads: int, rank: int, world: int):
    """Slice of one interval's reads binned by `rank`: [lo, hi) relative to the interval start."""
    return (interval_reads * rank) // world, (interval_reads * (rank + 1)) // world


def interval_slice(scaling: str, t: int, interval: int, rank: int, world: int):
    """(first global read, count) of the part of sketching interval `t` that `rank` bins.

    "strong" is SURVEY.md §8(e) / the reference's rule (pipeline/sketch.go:211-215: a flush every `interval` reads of
    the GLOBAL stream): interval t = global reads [t*I, (t+1)*I), rank g takes the contiguous slice
    [t*I + g*I/G, t*I + (g+1)*I/G) — the sketch is the one a single GPU computes with the same interval.
    "weak" keeps the per-rank work fixed instead: the global interval is G*I reads and rank g takes its g-th block of
    I — the sketch of a single-GPU run with interval G*I."""
    if scaling == "strong":
        lo, hi = read_shard(interval, rank, world)
        return t * interval + lo, hi - lo
    if scaling == "weak":
        return (t * world + rank) * interval, interval
    raise ValueError("scaling must be 'strong' or 'weak'")


def batch_share(step: int, batch: int, interval: int, rank: int, world: int):
    """Whole intervals of ONE batch of `batch` intervals shared among the ranks (hulk_bin_reads_device_at): rank g bins
    the intervals [g*batch/G, (g+1)*batch/G) of batch `step`.  Returns (first global read, number of reads, first spectrum
    of the batch); needs batch % world == 0.  (hulk_step_sharded gives every rank a whole batch instead: step_share.)"""
    if batch % world:
        raise ValueError("batch_share needs the batch size to be a multiple of the number of ranks")
    per = batch // world
    return (step * batch + rank * per) * interval, per * interval, rank * per


def num_steps(total_reads: int, batch: int, interval: int, world: int) -> int:
    """Steps of hulk_step_sharded a stream of `total_reads` reads takes (a step = world * batch intervals)."""
    per_step = world * batch * interval
    return (total_reads + per_step - 1) // per_step


def step_share(step: int, batch: int, interval: int, rank: int, world: int, total_reads=None):
    """hulk_step_sharded's layout: step `step` covers the global intervals [step*G*T, (step+1)*G*T) (T = `batch` =
    hulk_batch_size), rank g holds [step*G*T + g*T, step*G*T + (g+1)*T).  Returns (first global read, number of reads,
    step_intervals): the rank's contiguous chunk and the number of intervals of the global stream in this step — the same
    on every rank.  With `total_reads` the stream ends there: the last step is ragged (a rank may hold fewer intervals
    or none, the stream's last interval may be partial — the reference's EOF flush, pipeline/sketch.go:219-221)."""
    per_step = world * batch * interval
    lo = step * per_step
    hi = lo + per_step if total_reads is None else min(lo + per_step, total_reads)
    if hi <= lo:
        return lo, 0, 0
    step_intervals = (hi - lo + interval - 1) // interval
    first = min(lo + rank * batch * interval, hi)
    last = min(first + batch * interval, hi)
    return first, last - first, step_intervals


def gloo_exchange(dist, group=None):
    """The exchange function hulk_comm_init_host wants, over a torch.distributed process group on HOST tensors (gloo):
    exchange(op, send, recv) with numpy uint8 views of the library's pinned staging."""
    import torch

    def exchange(op, send, recv):
        if op == 0:                                           # HULK_XCHG_ALLGATHER: world * bytes, rank order
            world = dist.get_world_size(group)
            t = torch.from_numpy(send)
            out = torch.from_numpy(recv)
            dist.all_gather(list(out.view(world, -1).unbind(0)), t, group=group)
        elif op == 1:                                         # HULK_XCHG_ALLREDUCE_U32: uint32 sum (int32 adds wrap alike)
            t = torch.from_numpy(send.view(np.int32).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            recv.view(np.int32)[:] = t.numpy()
        else:
            raise ValueError(f"unknown exchange op {op}")
    return exchange
