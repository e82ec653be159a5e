"""Host ingest (include/hulk_hip.h, "host ingest"): the reference's DataStreamer.Run +
FastqHandler.Run (src/pipeline/sketch.go:40-161) in native code.

`parse_files` needs no GPU: it returns the reads the reference would hand to theBoss.AddSeq.
`GpuSketcher.sketch_files` (sketcher.py) is the same parser feeding the GPU path.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import BATCH_FN, HulkError, IngestOpts, IngestStats


def make_opts(opts):
    """dict of hulk_ingest_opts fields (parser_threads, gz_threads, file_readers, flags, block_bytes, gz_chunk_bytes) -> struct"""
    return IngestOpts(**opts) if opts else None


def _path_array(paths):
    enc = [p.encode() if isinstance(p, str) else bytes(p) for p in paths]
    arr = (ctypes.c_char_p * max(len(enc), 1))(*enc) if enc else (ctypes.c_char_p * 1)()
    return arr, len(enc)


def parse_files(paths, fasta=False, threads=0, collect=True, opts=None):
    """-> (bases uint8[], offsets uint64[n+1], stats dict).  paths == [] reads STDIN.
    collect=False parses without handing the batches over (timing aid): empty arrays + stats.
    opts: dict of hulk_ingest_opts fields for this run (hulk_parse_files_opts)."""
    L = _lib.load()
    arr, n = _path_array(paths)
    chunks, lens = [], []

    def on_batch(_user, bases, offsets, n_reads):
        off = np.ctypeslib.as_array(offsets, shape=(n_reads + 1,)).copy()
        total = int(off[-1])
        chunks.append(np.ctypeslib.as_array(bases, shape=(max(total, 1),))[:total].copy())
        lens.append(np.diff(off))
        return 0

    cb = BATCH_FN(on_batch) if collect else ctypes.cast(None, BATCH_FN)
    st = IngestStats()
    err = ctypes.create_string_buffer(1024)
    if opts:
        o = make_opts(dict({"parser_threads": threads}, **opts))
        rc = L.hulk_parse_files_opts(arr, n, 1 if fasta else 0, ctypes.byref(o), cb, None, ctypes.byref(st), err, 1024)
    else:
        rc = L.hulk_parse_files(arr, n, 1 if fasta else 0, threads, cb, None, ctypes.byref(st), err, 1024)
    if rc != 0:
        raise HulkError(rc, err.value.decode("latin-1") or L.hulk_strerror(rc).decode())
    bases = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    ln = np.concatenate(lens) if lens else np.zeros(0, dtype=np.uint64)
    offsets = np.zeros(len(ln) + 1, dtype=np.uint64)
    np.cumsum(ln, out=offsets[1:])
    return bases, offsets, stats_dict(st)


def stats_dict(st):
    return {"n_seqs": int(st.n_seqs), "total_len": int(st.total_len), "n_lines": int(st.n_lines),
            "bytes_in": int(st.bytes_in), "seconds": float(st.seconds)}
