"""`hulk smash` on the GPU: pairwise similarity matrix of a directory of HULK sketches.

Reference: cmd/smash.go:60-226 (parameter checks, CollectJSONs, makeMatrix), HULKdata.GetDistance
(src/sketchio/sketchio.go:259-306), distances.GetDistance/GetWJD (src/distances/distances.go).
The N x N x S comparison runs in libhulkhip (hulk_smash); this module only loads, orders and writes.
"""
import fnmatch
import glob
import os

import numpy as np

from . import _lib
from ._lib import HulkError
from .sketchio import load_hulk_data

AVAIL_METRICS = ["jaccard", "weightedjaccard"]          # cmd/smash.go:30 (the others are unreachable from the CLI)
AVAIL_ALGORITHMS = ["histosketch", "kmv", "khf"]        # sketchio.go:17


def collect_jsons(sketch_dir, recursive=False):
    """helpers.CollectJSONs (src/helpers/helpers.go:168-208)."""
    if not sketch_dir.endswith("/"):
        sketch_dir += "/"
    if recursive:
        found = []
        for root, _, files in os.walk(sketch_dir):
            found += [os.path.join(root, f) for f in sorted(files) if fnmatch.fnmatch(f, "*.json")]
    else:
        found = glob.glob(sketch_dir + "*.json")
    if not found:
        raise HulkError(-30, f"no JSON files found in supplied directory: {sketch_dir}\n")
    return found


def distance_matrix(mins, weights, metric="jaccard", device=0, timing=None):
    """distances[s, q] = GetDistance(subject s, query q) for all pairs, computed on the GPU.
    timing: a dict that receives "kernel_ms" (k_smash alone, HIP events; hulk_smash_ex)."""
    mins = np.ascontiguousarray(mins, dtype=np.uint64)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    if mins.shape != weights.shape or mins.ndim != 2:
        raise ValueError("mins/weights must be [n_sketches][sketch_size]")
    n, s = mins.shape
    out = np.zeros((n, n), dtype=np.float64)
    L = _lib.load()
    import ctypes
    kms = ctypes.c_double(0.0)
    rc = L.hulk_smash_ex(device, mins.ctypes.data, weights.ctypes.data, n, s,
                         1 if metric == "weightedjaccard" else 0, out.ctypes.data,
                         ctypes.byref(kms) if timing is not None else None)
    if timing is not None:
        timing["kernel_ms"] = kms.value
    if rc != 0:
        raise HulkError(rc, L.hulk_last_error(None).decode())
    return out


def go_format_f2(v: float) -> str:
    """strconv.FormatFloat(v, 'f', 2, 64)"""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "+Inf" if v > 0 else "-Inf"
    return f"{v:.2f}"


def go_csv_field(f: str) -> str:
    """encoding/csv Writer: quote a field only when it needs it (fieldNeedsQuotes)."""
    need = f == "\\." or any(ch in f for ch in ',"\r\n') or (f != "" and f[0] in " \t")
    return '"' + f.replace('"', '""') + '"' if need else f


def find_sketch(data, ksize, algo, path):
    """HULKdata.FindSketch (sketchio.go:198-257) for histosketch signatures."""
    if algo not in AVAIL_ALGORITHMS:
        raise HulkError(-30, f"specified algorithm ({algo}) not found in the supplied sketch: {data.filename}\n")
    sigs = [hs for a, hs in data.signatures if a == algo]
    if not sigs:
        raise HulkError(-30, f"no sketches were produced using the {algo} algorithm in file: {data.filename}\n")
    hit = [hs for hs in sigs if hs.ksize == ksize]
    if len(hit) > 1:
        raise HulkError(-30, f"found {len(hit)} possible duplicate sketches in the supplied sketch file: {data.filename}\n")
    if not hit:
        raise HulkError(-30, f"specified k-mer size ({ksize}) not found in the supplied sketch file: {data.filename}\n")
    return hit[0]


def _paths(files):
    import ctypes
    enc = [os.fsencode(f) for f in files]
    return (ctypes.c_char_p * max(len(enc), 1))(*enc), len(enc)


def load_sketches(files, ksize=21, algo="histosketch", threads=0):
    """LoadHULKdata + FindSketch for every file, in native code on `threads` host threads (hulk_load_sketches; no GPU needed):
    -> (ordering, mins[n][S], weights[n][S], banner labels), ordering = the sorted paths.  Raises HulkError with the reference's text."""
    import ctypes
    L = _lib.load()
    arr, n = _paths(files)
    h = ctypes.c_void_p()
    err = ctypes.create_string_buffer(4096)
    rc = L.hulk_load_sketches(arr, n, ksize, algo.encode(), threads, ctypes.byref(h), err, len(err))
    if rc != 0:
        raise HulkError(rc, err.value.decode("utf-8", "replace"))
    try:
        cn, cs = ctypes.c_uint32(), ctypes.c_uint32()
        L.hulk_sketch_set_info(h, ctypes.byref(cn), ctypes.byref(cs))
        n, S = cn.value, cs.value
        mins = np.ctypeslib.as_array(ctypes.cast(L.hulk_sketch_set_mins(h), ctypes.POINTER(ctypes.c_uint64)), shape=(n * max(S, 1),))[:n * S].reshape(n, S).copy()
        weights = np.ctypeslib.as_array(ctypes.cast(L.hulk_sketch_set_weights(h), ctypes.POINTER(ctypes.c_double)), shape=(n * max(S, 1),))[:n * S].reshape(n, S).copy()
        ordering = [os.fsdecode(L.hulk_sketch_set_path(h, i)) for i in range(n)]
        banners = [L.hulk_sketch_set_banner(h, i).decode("utf-8", "replace") for i in range(n)]
    finally:
        L.hulk_sketch_set_free(h)
    return ordering, mins, weights, banners


def smash(sketch_dir, out_file, ksize=21, algo="histosketch", metric="jaccard", recursive=False, device=0,
          banner_matrix=False, stages=None, threads=0):
    """runSmash + makeMatrix (cmd/smash.go:60-226), native from the file names on (hulk_smash_files): the JSON files are parsed
    and MD5-verified on `threads` host threads, the N x N x S comparison runs on the GPU, the CSV is written by the library.
    Writes <out_file>.hulk-matrix.csv (and, banner_matrix, <out_file>.banner-matrix.csv: makeBannerMatrix, cmd/smash.go:229-261 —
    the reference iterates a Go map, here the sorted file order is used) and returns (ordering, distances).
    stages: a dict that receives the seconds of "load" (JSON + MD5 check), "matrix", "csv" and "kernel_ms"."""
    import ctypes
    if metric not in AVAIL_METRICS:
        raise HulkError(-30, f"supplied distance metric is not available: {metric}\nplease select one of the following: {AVAIL_METRICS}")
    if algo not in AVAIL_ALGORITHMS:
        raise HulkError(-30, f"supplied algorithm not available: {algo}\nplease select one of the following: {AVAIL_ALGORITHMS}")
    files = collect_jsons(sketch_dir, recursive)
    od = os.path.dirname(out_file)
    if od and od != "." and not os.path.exists(od):
        os.makedirs(od, mode=0o700)
    L = _lib.load()
    arr, n = _paths(files)
    n_unique = len(set(files))
    dist = np.zeros((n_unique, n_unique), dtype=np.float64)
    st = _lib.SmashStats()
    err = ctypes.create_string_buffer(4096)
    rc = L.hulk_smash_files(device, arr, n, ksize, algo.encode(), metric.encode(), threads, os.fsencode(out_file + ".hulk-matrix.csv"),
                            os.fsencode(out_file + ".banner-matrix.csv") if banner_matrix else None, dist.ctypes.data,
                            ctypes.byref(st), err, len(err))
    if rc != 0:
        raise HulkError(rc, err.value.decode("utf-8", "replace"))
    if stages is not None:
        stages.update(load=st.seconds_load, matrix=st.seconds_matrix, csv=st.seconds_csv, kernel_ms=st.kernel_ms)
    return sorted(set(files)), dist


def smash_python(sketch_dir, out_file, ksize=21, algo="histosketch", metric="jaccard", recursive=False, device=0,
          banner_matrix=False, stages=None):
    """The same run with the load, the ordering and the CSV in Python (json + hashlib; the form `smash` had until round 6, kept as
    the comparator of the native one in tests/).  runSmash + makeMatrix: writes <out_file>.hulk-matrix.csv and returns (ordering, distances).
    banner_matrix: also <out_file>.banner-matrix.csv (makeBannerMatrix, cmd/smash.go:229-261): one line
    per sketch = its mins + the banner label; the reference iterates a Go map (random order), here the
    sorted file order is used.  stages: a dict that receives the seconds of "load" (JSON + MD5 check), "matrix", "csv"."""
    import time
    t_start = time.perf_counter()
    if metric not in AVAIL_METRICS:
        raise HulkError(-30, f"supplied distance metric is not available: {metric}\nplease select one of the following: {AVAIL_METRICS}")
    if algo not in AVAIL_ALGORITHMS:
        raise HulkError(-30, f"supplied algorithm not available: {algo}\nplease select one of the following: {AVAIL_ALGORITHMS}")
    files = collect_jsons(sketch_dir, recursive)
    loaded = {}
    for f in files:
        try:
            loaded[f] = load_hulk_data(f)
        except ValueError as e:
            raise HulkError(-30, str(e))
    if len(loaded) < 2:
        raise HulkError(-30, f"{len(loaded)} sketches found in the supplied directory, HULK needs at least 2 to smash!\n")
    ordering = sorted(loaded)                                   # sort.Strings (byte order)
    sk = [find_sketch(loaded[f], ksize, algo, f) for f in ordering]
    size = len(sk[0].mins)
    for a in sk:
        if len(a.mins) != size:
            raise HulkError(-30, f"sketch length mismatch: {size} vs {len(a.mins)}\n")
    mins = np.stack([a.mins for a in sk])
    if algo == "histosketch":
        weights = np.stack([a.weights for a in sk])
    elif metric == "weightedjaccard":                           # sketchio.go:287-293
        raise HulkError(-30, "weighted jaccard is only supported for histosketches")
    else:
        weights = np.zeros(mins.shape)                          # MinHash signatures carry no weights (khf.go:12-16)
    t_loaded = time.perf_counter()
    dist = distance_matrix(mins, weights, metric, device)
    t_matrix = time.perf_counter()
    od = os.path.dirname(out_file)
    if od and od != "." and not os.path.exists(od):
        os.makedirs(od, mode=0o700)
    with open(out_file + ".hulk-matrix.csv", "w", encoding="utf-8", newline="") as fh:
        fh.write(",".join(go_csv_field(f) for f in ordering) + "\n")
        for row in dist:
            v = 100 - (row * 100)                                  # (elementwise IEEE double arithmetic: the Go expression)
            if np.isfinite(v).all():                               # FormatFloat(v, 'f', 2, 64) == "%.2f" for finite values
                fh.write(",".join(["%.2f" % x for x in v.tolist()]) + "\n")
            else:
                fh.write(",".join(go_format_f2(x) for x in v.tolist()) + "\n")
    if banner_matrix:
        with open(out_file + ".banner-matrix.csv", "w", encoding="utf-8", newline="") as fh:
            for f, a in zip(ordering, sk):
                fh.write(",".join([str(int(v)) for v in a.mins] + [go_csv_field(loaded[f].banner_label)]) + "\n")
    if stages is not None:
        stages.update(load=t_loaded - t_start, matrix=t_matrix - t_loaded, csv=time.perf_counter() - t_matrix)
    return ordering, dist
