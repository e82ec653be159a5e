"""Multi-GPU sharding of the sketch path: one process per GPU.

Per interval t (global reads [tI, (t+1)I)):
  1. rank g bins its contiguous slice [tI + gI/G, tI + (g+1)I/G) into a private uint32 histogram
  2. ONE exchange: all-reduce(sum) of the histogram (k^4 uint32; RCCL over xGMI on GPUs) — every
     rank needs every bin because count-min collisions couple bins.  Intervals are processed in
     batches of T (hulk_batch_size): T spectra are merged by ONE all-reduce of T*k^4 uint32, which
     turns T latency-bound 0.8 MB messages into one bandwidth-bound message
  3. the count-min update is replicated (cheap, deterministic); the CWS update is slot-sharded:
     rank g owns sketch slots [gS/G, (g+1)S/G) and only that slice of the CWS tables
  4. at EOF one all-gather of the per-rank (mins, weights) slices

The class is engine-agnostic: the product engine is `GpuSketcher` (libhulkhip); the CPU test
suite drives the same logic with a test double over gloo (tests/test_distributed_cpu.py).
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
    """The other way to split the SAME global stream with the SAME interval (so: the same sketch as "strong"): inside a
    batch of `batch` consecutive intervals rank g bins the WHOLE intervals [g*batch/G, (g+1)*batch/G) — one contiguous
    chunk of reads per batch, batch/G spectra to build instead of `batch` slices, and the all-reduce over the ring is a
    gather.  Returns (first global read, number of reads, first spectrum of the batch); needs batch % world == 0."""
    if batch % world:
        raise ValueError("batch_share needs the batch size to be a multiple of the number of ranks")
    per = batch // world
    return (step * batch + rank * per) * interval, per * interval, rank * per


class ShardedSketcher:
    """Drives one rank of a G-rank run.

    engine must provide: bin_reads(first_read_in_interval_slice...) is left to the caller; this
    class needs only
        engine.histogram_tensor() -> tensor viewing the engine's histogram (summed in place)
        engine.flush()
        engine.finish()
        engine.sketch() -> (mins uint64[S], weights float64[S])  (own slots filled)
    and a torch.distributed-like module `dist` (all_reduce, all_gather_object / all_gather).
    """

    def __init__(self, engine, sketch_size, rank, world, dist=None, group=None):
        self.engine, self.S, self.rank, self.world = engine, sketch_size, rank, world
        self.dist, self.group = dist, group
        self.slot_begin, self.slot_count = slot_shard(sketch_size, rank, world)

    def end_interval(self):
        """Steps 2+3 for the interval whose reads the caller has just binned."""
        if self.world > 1:
            h = self.engine.histogram_tensor()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
        self.engine.flush()

    def finish(self):
        if self.world > 1:
            h = self.engine.histogram_tensor()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
        self.engine.finish()

    def gather_sketch(self):
        """Step 4: full (mins, weights) on every rank."""
        mins, weights = self.engine.sketch()
        if self.world == 1:
            return mins, weights
        import torch
        lo, n = self.slot_begin, self.slot_count
        # fixed-size payload per rank: pad to the largest shard
        cap = max(slot_shard(self.S, r, self.world)[1] for r in range(self.world))
        pay = np.zeros(2 * cap, dtype=np.int64)
        pay[:n] = mins[lo:lo + n].view(np.int64)
        pay[cap:cap + n] = weights[lo:lo + n].view(np.int64)
        t = torch.from_numpy(pay)
        dev = getattr(self.engine, "collective_device", None)
        if dev is not None:
            t = t.to(dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        full_m = np.zeros(self.S, dtype=np.uint64)
        full_w = np.zeros(self.S, dtype=np.float64)
        for r, o in enumerate(outs):
            b, c = slot_shard(self.S, r, self.world)
            a = o.cpu().numpy()
            full_m[b:b + c] = a[:c].view(np.uint64)
            full_w[b:b + c] = a[cap:cap + c].view(np.float64)
        return full_m, full_w


class GpuEngine:
    """Adapter: GpuSketcher + a torch view of its device histogram for the collective."""

    def __init__(self, sketcher, device, n_spectra=1):
        import torch
        self.sk = sketcher
        self.collective_device = torch.device(device)
        self.n_spectra = n_spectra          # spectra (intervals) merged per collective
        self._views = {}                    # the library alternates between two spectrum rings

    def histogram_tensor(self):
        """torch view of the spectra the NEXT flush will consume (int32: counts < 2^31, sum bit-identical)."""
        import torch
        ptr = self.sk.histogram_device_ptr()
        t = self._views.get(ptr)
        if t is None:
            nb = self.sk.num_bins * self.n_spectra

            class _View:  # __cuda_array_interface__ v2
                __cuda_array_interface__ = {"shape": (nb,), "typestr": "<i4", "data": (ptr, False),
                                            "version": 2, "strides": None}
            t = torch.as_tensor(_View(), device=self.collective_device)
            self._views[ptr] = t
        return t

    def flush(self): self.sk.flush_batch(self.n_spectra)
    def finish(self): self.sk.finish()
    def sketch(self): return self.sk.sketch()
