"""Host-side mirror of the reference's sketching seam, over the libhulkhip C ABI.

Reference objects mirrored here (will-rowe/hulk v1.0.0):
  * `theBoss` (src/pipeline/boss.go:10-41): AddSeq / Flush / StopWork / GetMinimizerCount
  * `histosketch.HistoSketch` (src/histosketch/histosketch.go:36-47): the exported fields
    Sketch (`mins`), SketchWeights (`weights`), KmerSize, SketchSize, Dimensions,
    ApplyConceptDrift — consumed by sketchio.HULKdata.Add / WriteJSON
  * the interval rule of SeqMinimizer.Run (src/pipeline/sketch.go:196-224)

Every numeric step runs on the GPU inside libhulkhip.so; this module only marshals buffers.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import HulkError, HulkParams


def spectrum_size(k: int) -> int:
    """int32(helpers.Pow(k, 4)) — cmd/sketch.go:118."""
    v = (k ** 4) & 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


class HistoSketch:
    """The result object: field names follow histosketch.HistoSketch's JSON tags."""
    algorithm = "histosketch"

    def __init__(self, ksize, mins, weights, num_histogram_bins, concept_drift):
        self.ksize = int(ksize)
        self.mins = np.asarray(mins, dtype=np.uint64)
        self.weights = np.asarray(weights, dtype=np.float64)
        self.num = len(self.mins)
        self.num_histogram_bins = int(num_histogram_bins)
        self.concept_drift = bool(concept_drift)
        self.md5sum = ""

    def get_sketch(self):
        return self.mins


class GpuSketcher:
    """boss + sketcher for one `hulk sketch` run on one GPU (or one rank's shard of it)."""

    def __init__(self, k=21, w=9, sketch_size=50, interval=0, decay_ratio=1.0, num_bins=0,
                 device=0, slot_begin=0, slot_count=0, cws_source=_lib.HULK_CWS_GO_COMPAT,
                 stream=None, flags=0, batch=0, work_lanes=0, host_copy_threads=0):
        """batch: sketching intervals per flush batch (0 = the library's default, 16); work_lanes: 2 (the default) = consecutive
        batches are binned on two alternating work streams, 1 = one; host_copy_threads: see hulk_params (include/hulk_hip.h).
        stream: run on the caller's hipStream_t — the context then joins its second lane into that stream after every call."""
        self._L = _lib.load()
        self._ctx = ctypes.c_void_p()
        p = HulkParams(k=k, w=w, sketch_size=sketch_size, num_bins=num_bins,
                       decay_ratio=decay_ratio, interval=interval, device=device,
                       slot_begin=slot_begin, slot_count=slot_count, cws_source=cws_source, flags=flags,
                       batch=batch, work_lanes=work_lanes, host_copy_threads=host_copy_threads)
        rc = self._L.hulk_create(ctypes.byref(p), ctypes.byref(self._ctx))
        if rc != 0:
            self._ctx = None
            raise HulkError(rc, self._L.hulk_last_error(None).decode())
        self.k, self.w, self.sketch_size = k, w, sketch_size
        self.interval, self.decay_ratio = interval, decay_ratio
        self.num_bins = num_bins if num_bins else spectrum_size(k)
        self.slot_begin = slot_begin
        self.slot_count = slot_count if slot_count else sketch_size
        if stream is not None:
            self.set_stream(stream)

    # ---- lifetime
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.hulk_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def _chk(self, rc):
        if rc != 0:
            raise HulkError(rc, self._L.hulk_last_error(self._ctx).decode())

    def set_stream(self, stream_handle):
        """Run on the caller's hipStream_t (0/None = the HIP null stream, torch's default)."""
        self._chk(self._L.hulk_set_stream(self._ctx, ctypes.c_void_p(stream_handle or 0)))

    def set_private_stream(self):
        self._chk(self._L.hulk_set_private_stream(self._ctx))

    # ---- theBoss
    def add_seq(self, seq: bytes):
        """theBoss.AddSeq (boss.go:24-26) for a single sequence."""
        arr = np.frombuffer(seq, dtype=np.uint8)
        self.add_reads(arr, np.array([0, len(arr)], dtype=np.uint64))

    def add_reads(self, bases, offsets):
        """AddSeq for a batch held in host memory: read i = bases[offsets[i]:offsets[i+1]]."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(self._L.hulk_add_reads(self._ctx, bases.ctypes.data, offsets.ctypes.data,
                                         len(offsets) - 1))

    def scan_stats(self):
        """(tiles of the CWS table the scans read, tiles they covered) — see hulk_get_scan_stats."""
        import ctypes
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._chk(self._L.hulk_get_scan_stats(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def sketch_files(self, paths, fasta=False, threads=0, opts=None):
        """DataStreamer + FastqHandler + the AddSeq loop (pipeline/sketch.go:40-217) in native code:
        parse the inputs ([] = STDIN, *.gz gunzipped) and add every read.  Returns the ingest stats.
        opts: dict of hulk_ingest_opts fields for this run."""
        import ctypes
        from ._lib import IngestStats
        from .ingest import _path_array, make_opts, stats_dict
        arr, n = _path_array(paths)
        st = IngestStats()
        if opts:
            o = make_opts(dict({"parser_threads": threads}, **opts))
            self._chk(self._L.hulk_sketch_files_opts(self._ctx, arr, n, 1 if fasta else 0, ctypes.byref(o), ctypes.byref(st)))
            return stats_dict(st)
        self._chk(self._L.hulk_sketch_files(self._ctx, arr, n, 1 if fasta else 0, threads, ctypes.byref(st)))
        return stats_dict(st)

    def add_reads_device(self, bases_ptr, offsets_ptr, n_reads, max_read_len, bases_bytes):
        """AddSeq for a batch already resident in HBM (raw device pointers)."""
        self._chk(self._L.hulk_add_reads_device(self._ctx, bases_ptr, offsets_ptr, n_reads,
                                                max_read_len, bases_bytes))

    def bin_reads_device(self, bases_ptr, offsets_ptr, n_reads, max_read_len, bases_bytes,
                         reads_per_spectrum=0, first_spectrum=0):
        """Multi-GPU step 1: bin this rank's reads, no interval rule (see hulk_hip.h); first_spectrum: the spectrum of
        the batch the first read belongs to (a rank that bins whole intervals of a batch)."""
        self._chk(self._L.hulk_bin_reads_device_at(self._ctx, bases_ptr, offsets_ptr, n_reads,
                                                   max_read_len, bases_bytes, reads_per_spectrum, first_spectrum))

    def flush_batch(self, n_spectra, after_stream=None):
        """Flush n_spectra spectra; with after_stream (a hipStream_t handle) the flush waits for that
        stream (the collective's) instead of the work stream."""
        if after_stream is None:
            self._chk(self._L.hulk_flush_batch(self._ctx, n_spectra))
        else:
            self._chk(self._L.hulk_flush_batch_after(self._ctx, n_spectra, ctypes.c_void_p(after_stream)))

    # ---- multi-GPU: the exchange inside the library (include/hulk_hip.h)
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: called by one rank, handed to the others by the host."""
        L = _lib.load()
        buf = ctypes.create_string_buffer(_lib.HULK_UNIQUE_ID_BYTES)
        rc = L.hulk_comm_unique_id(buf)
        if rc != 0:
            raise HulkError(rc, L.hulk_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """RCCL communicator of this rank (collective call)."""
        if len(unique_id) != _lib.HULK_UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be HULK_UNIQUE_ID_BYTES long")
        self._chk(self._L.hulk_comm_init(self._ctx, ctypes.c_char_p(unique_id), rank, world))
        self.rank, self.world = rank, world

    def comm_init_host(self, rank: int, world: int, exchange):
        """The same protocol over a transport of the host: exchange(op, send: np.uint8[bytes], recv: np.uint8[...]) fills
        `recv` (op 0: all-gather, world * bytes in rank order; op 1: uint32 sum, bytes)."""
        def thunk(_user, op, send, recv, nbytes):
            try:
                n = int(nbytes)
                s_ = np.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_uint8)), shape=(n,))
                r_ = np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_uint8)),
                                           shape=(n * world if op == _lib.HULK_XCHG_ALLGATHER else n,))
                exchange(int(op), s_, r_)
                return 0
            except Exception as e:  # noqa: BLE001 — reported through the ABI's status code
                import sys
                sys.stderr.write(f"hulk_amd: exchange callback failed: {e!r}\n")
                return 1
        self._exchange_thunk = _lib.EXCHANGE_FN(thunk)           # keep the callback alive as long as the context
        self._chk(self._L.hulk_comm_init_host(self._ctx, rank, world, self._exchange_thunk, None))
        self.rank, self.world = rank, world

    def comm_init_loopback(self, rank: int, world: int):
        """Projection aid: one rank's share of a `world`-rank step on one GPU, no peers."""
        self._chk(self._L.hulk_comm_init_loopback(self._ctx, rank, world))
        self.rank, self.world = rank, world

    def step_sharded(self, bases_ptr, offsets_ptr, n_reads, max_read_len, bases_bytes, step_intervals):
        """One step of this rank: its whole intervals of the step -> exchange -> flush (hulk_step_sharded)."""
        self._chk(self._L.hulk_step_sharded(self._ctx, bases_ptr, offsets_ptr, n_reads, max_read_len, bases_bytes,
                                            step_intervals))

    def step_sharded_host(self, bases, offsets, step_intervals):
        """hulk_step_sharded for reads held in host memory (numpy): read i = bases[offsets[i]:offsets[i+1]]."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(self._L.hulk_step_sharded_host(self._ctx, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                                 step_intervals))

    def step_sliced(self, bases_ptr, offsets_ptr, n_reads, max_read_len, bases_bytes, reads_per_spectrum, n_spectra):
        """SURVEY.md 8(e) to the letter: a slice of every interval per rank, one all-reduce of the spectra, flush."""
        self._chk(self._L.hulk_step_sliced(self._ctx, bases_ptr, offsets_ptr, n_reads, max_read_len, bases_bytes,
                                           reads_per_spectrum, n_spectra))

    def gather_sketch(self):
        """(mins, weights) of the whole sketch on every rank."""
        mins = np.zeros(self.sketch_size, dtype=np.uint64)
        weights = np.zeros(self.sketch_size, dtype=np.float64)
        self._chk(self._L.hulk_gather_sketch(self._ctx, mins.ctypes.data, weights.ctypes.data))
        return mins, weights

    def comm_stats(self):
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(self._L.hulk_get_comm_stats(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        h, v = ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(self._L.hulk_get_comm_health(self._ctx, ctypes.byref(h), ctypes.byref(v)))
        return {"steps_delta": a.value, "steps_full": b.value, "bytes_received": c.value,
                "headers_refetched": h.value, "void_blocks": v.value}

    def _need_experiments(self):
        if not _lib.is_experiments_build():
            raise RuntimeError("test hook of the profiling build: run with HULK_LIB=exp (make -C hulk_amd/csrc EXPERIMENTS=1)")

    def debug_read(self, what: int):
        """Test hook (hulk_debug_read): HULK_DEBUG_TILEMIN -> float32 array, HULK_DEBUG_SCANMAP -> uint64 array."""
        self._need_experiments()
        n = ctypes.c_uint64(0)
        buf = np.zeros(1, dtype=np.uint8)
        self._L.hulk_debug_read(self._ctx, what, buf.ctypes.data, ctypes.byref(n))       # (too small: reports the size)
        buf = np.zeros(n.value, dtype=np.uint8)
        self._chk(self._L.hulk_debug_read(self._ctx, what, buf.ctypes.data, ctypes.byref(n)))
        return buf.view(np.float32 if what == _lib.HULK_DEBUG_TILEMIN else np.uint64)

    def debug_inject(self, what: int, step: int):
        """Test hook (hulk_debug_inject): make this rank's header block of `step` void / its host staging late."""
        self._need_experiments()
        self._chk(self._L.hulk_debug_inject(self._ctx, what, step))

    @property
    def batch_size(self):
        return self._L.hulk_batch_size(self._ctx)

    def histogram_device_ptr(self):
        return self._L.hulk_histogram_device(self._ctx)

    def add_histogram(self, hist):
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        if len(hist) != self.num_bins:
            raise ValueError("histogram length != num_bins")
        self._chk(self._L.hulk_add_histogram(self._ctx, hist.ctypes.data))

    def set_cws_tables(self, r, c, b):
        r, c, b = (np.ascontiguousarray(x, dtype=np.float64) for x in (r, c, b))
        self._chk(self._L.hulk_set_cws_tables(self._ctx, r.ctypes.data, c.ctypes.data, b.ctypes.data))

    def flush(self):
        """theBoss.Flush (boss.go:34-36)."""
        self._chk(self._L.hulk_flush(self._ctx))

    def synchronize(self):
        """Wait for everything queued so far (copies, binning kernels, flushes)."""
        self._chk(self._L.hulk_synchronize(self._ctx))

    def stop_work(self):
        """Final flush + theBoss.StopWork (pipeline/sketch.go:219-224)."""
        self._chk(self._L.hulk_finish(self._ctx))

    finish = stop_work

    def get_minimizer_count(self):
        return self.counters()["n_minimizers"]

    # ---- outputs
    def counters(self):
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(self._L.hulk_get_counters(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"n_reads": a.value, "n_minimizers": b.value, "total_len": c.value}

    def sketch(self):
        mins = np.zeros(self.sketch_size, dtype=np.uint64)
        weights = np.zeros(self.sketch_size, dtype=np.float64)
        self._chk(self._L.hulk_get_sketch(self._ctx, mins.ctypes.data, weights.ctypes.data))
        return mins, weights

    def histosketch(self) -> HistoSketch:
        mins, weights = self.sketch()
        return HistoSketch(self.k, mins, weights, self.num_bins, self.decay_ratio != 1.0)

    def histogram(self):
        h = np.zeros(self.num_bins, dtype=np.uint32)
        self._chk(self._L.hulk_get_histogram(self._ctx, h.ctypes.data))
        return h

    def cms(self):
        a = np.zeros(7 * 2000, dtype=np.float64)
        self._chk(self._L.hulk_get_cms(self._ctx, a.ctypes.data))
        return a.reshape(7, 2000)

    def cws_tables(self):
        n = self.slot_count * self.num_bins
        r, c, b = np.empty(n), np.empty(n), np.empty(n)
        self._chk(self._L.hulk_get_cws_tables(self._ctx, r.ctypes.data, c.ctypes.data, b.ctypes.data))
        shp = (self.slot_count, self.num_bins)
        return r.reshape(shp), c.reshape(shp), b.reshape(shp)

    def selftest_reciprocal(self):
        n = ctypes.c_uint64()
        self._chk(self._L.hulk_selftest_reciprocal(self._ctx, ctypes.byref(n)))
        return n.value

    def device_checks(self):
        """What hulk_create verified on the device: {"lds_order_ok": ..., "cms_chain_form": ...} (hulk_get_device_checks)."""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        self._chk(self._L.hulk_get_device_checks(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return {"lds_order_ok": bool(a.value), "cms_chain_form": bool(b.value)}

    def profile_table(self):
        """{kernel: (launches, total_ms)} of every launch since hulk_set_profiling(32) (hulk_get_profile_table; clears the log)."""
        buf = ctypes.create_string_buffer(1 << 16)
        self._chk(self._L.hulk_get_profile_table(self._ctx, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            k, n, ms = line.split("\t")
            out[k] = (int(n), float(ms))
        return out

    def set_profiling(self, on=True):
        self._chk(self._L.hulk_set_profiling(self._ctx, int(on)))

    def get_profile(self, kernel="k_cws_scan"):
        n, ms = ctypes.c_uint64(), ctypes.c_double()
        self._chk(self._L.hulk_get_profile(self._ctx, kernel.encode(), ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value
