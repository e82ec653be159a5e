"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under hulk_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborc.so")

ERRORS = {
    -1: "w must be: 0 < w < 257", -2: "k size must be: 0 < k < 32",
    -3: "sequence length must be > 0", -4: "sequence length must be >= w + k - 1",
    -5: "not used yet", -6: "histosketching only supports k <= 31",
    -7: "decay ratio must be between 0.0 and 1.0", -8: "histogram must have at least 2 bins",
    -9: "negative value used for number of k-mer spectrum bins", -10: "no sequences received",
    -20: "allocation failure",
}


def build(force=False):
    src = os.path.join(_HERE, "hulk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p, u32, i32, u64, i64, dbl, vp = (ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int32,
                                             ctypes.c_uint64, ctypes.c_int64, ctypes.c_double,
                                             ctypes.c_void_p)
        L.orc_nt4.restype = ctypes.c_uint8; L.orc_nt4.argtypes = [ctypes.c_uint8]
        L.orc_hash64.restype = u64; L.orc_hash64.argtypes = [u64, u64]
        L.orc_pow.restype = u64; L.orc_pow.argtypes = [u64, u64]
        L.orc_jump.restype = i32; L.orc_jump.argtypes = [u64, i64]
        L.orc_minimizers.restype = i32; L.orc_minimizers.argtypes = [u8p, i32, u32, u32, vp, i32]
        L.orc_gosrc_new.restype = vp; L.orc_gosrc_new.argtypes = [i64]
        L.orc_gosrc_free.argtypes = [vp]
        L.orc_gosrc_int63.restype = i64; L.orc_gosrc_int63.argtypes = [vp]
        L.orc_gosrc_uint64.restype = u64; L.orc_gosrc_uint64.argtypes = [vp]
        L.orc_gosrc_float64.restype = dbl; L.orc_gosrc_float64.argtypes = [vp]
        L.orc_go_gamma.restype = dbl; L.orc_go_gamma.argtypes = [vp, dbl, dbl]
        L.orc_go_uniform_range.restype = dbl; L.orc_go_uniform_range.argtypes = [vp, dbl, dbl]
        L.orc_set_gamma_variant.argtypes = [ctypes.c_int]
        L.orc_get_gamma_variant.restype = ctypes.c_int
        L.orc_cws_fill.restype = ctypes.c_int; L.orc_cws_fill.argtypes = [u32, i32, vp, vp, vp]
        L.orc_cms_geometry.argtypes = [vp, vp]
        L.orc_new.restype = ctypes.c_int; L.orc_new.argtypes = [u32, u32, u32, i32, dbl, u32, vp]
        L.orc_free.argtypes = [vp]
        L.orc_add_read.restype = ctypes.c_int; L.orc_add_read.argtypes = [vp, u8p, i32]
        L.orc_add_reads.restype = ctypes.c_int; L.orc_add_reads.argtypes = [vp, vp, vp, u64]
        L.orc_add_element.argtypes = [vp, u64, dbl]
        L.orc_add_histogram.restype = ctypes.c_int; L.orc_add_histogram.argtypes = [vp, vp]
        L.orc_flush.restype = ctypes.c_int; L.orc_flush.argtypes = [vp]
        L.orc_wipe.argtypes = [vp]
        L.orc_finish.restype = ctypes.c_int; L.orc_finish.argtypes = [vp]
        L.orc_get_sketch.argtypes = [vp, vp, vp]
        L.orc_get_histogram.argtypes = [vp, vp]
        L.orc_get_cms.argtypes = [vp, vp]
        L.orc_used_bins.restype = i32; L.orc_used_bins.argtypes = [vp]
        L.orc_num_bins.restype = i32; L.orc_num_bins.argtypes = [vp]
        L.orc_get_counters.argtypes = [vp] * 6
        for n in ("orc_cws_r", "orc_cws_c", "orc_cws_b"):
            getattr(L, n).restype = ctypes.POINTER(ctypes.c_double); getattr(L, n).argtypes = [vp]
        L.orc_smash_matrix.argtypes = [vp, vp, u32, u32, ctypes.c_int, vp]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(ERRORS.get(code, f"oracle error {code}"))
        self.code = code


def hash64(key, mask): return lib().orc_hash64(key, mask)
def jump(key, n): return lib().orc_jump(key & 0xFFFFFFFFFFFFFFFF, n)
def ipow(a, b): return lib().orc_pow(a, b)
def nt4(c): return lib().orc_nt4(c)


def minimizers(seq: bytes, k: int, w: int):
    """Distinct minimizer values of one read, first-emission order (reference: minimizer.go:59-204)."""
    out = np.zeros(max(len(seq), 1) + 1, dtype=np.uint64)
    n = lib().orc_minimizers(seq, len(seq), k, w, out.ctypes.data, len(out))
    if n < 0:
        raise OracleError(n)
    return out[:n].copy()


def set_gamma_variant(v):
    """0 = go_rng's recalled squeeze constant 4*exp(-0.5)/sqrt(2) (default), 1 = CPython's 1 + ln 4.5, 2 = no squeeze.
    Applies to tables generated afterwards (cws_tables, Sketcher).  All three give the same stream (see hulk_oracle.c)."""
    lib().orc_set_gamma_variant(int(v))


def cms_geometry():
    d, w = ctypes.c_uint32(), ctypes.c_uint32()
    lib().orc_cms_geometry(ctypes.byref(d), ctypes.byref(w))
    return d.value, w.value


def cws_tables(S, B):
    r = np.empty(S * B); c = np.empty(S * B); b = np.empty(S * B)
    lib().orc_cws_fill(S, B, r.ctypes.data, c.ctypes.data, b.ctypes.data)
    return r.reshape(S, B), c.reshape(S, B), b.reshape(S, B)


class GoRand:
    """Go math/rand source + go_rng generators on top."""
    def __init__(self, seed=1):
        self._p = lib().orc_gosrc_new(seed)
    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_gosrc_free(self._p); self._p = None
    def int63(self): return lib().orc_gosrc_int63(self._p)
    def uint64(self): return lib().orc_gosrc_uint64(self._p)
    def float64(self): return lib().orc_gosrc_float64(self._p)
    def gamma(self, alpha, beta): return lib().orc_go_gamma(self._p, alpha, beta)
    def uniform(self, a, b): return lib().orc_go_uniform_range(self._p, a, b)


class Sketcher:
    """The reference's boss + kmerspectrum + histosketch for one run (deterministic interval rule)."""
    def __init__(self, k=21, w=9, sketch_size=50, num_bins=0, decay_ratio=1.0, interval=0):
        p = ctypes.c_void_p()
        rc = lib().orc_new(k, w, sketch_size, num_bins, decay_ratio, interval, ctypes.byref(p))
        if rc != 0:
            raise OracleError(rc)
        self._p = p
        self.S = sketch_size
        self.B = lib().orc_num_bins(p)

    def close(self):
        if getattr(self, "_p", None):
            lib().orc_free(self._p); self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: the module globals may be gone already
            pass

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(rc)

    def add_read(self, seq: bytes): self._chk(lib().orc_add_read(self._p, seq, len(seq)))

    def add_reads(self, bases: np.ndarray, offsets: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(lib().orc_add_reads(self._p, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1))

    def add_histogram(self, hist):
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        assert len(hist) == self.B
        self._chk(lib().orc_add_histogram(self._p, hist.ctypes.data))

    def flush(self): self._chk(lib().orc_flush(self._p))
    def wipe(self): lib().orc_wipe(self._p)
    def finish(self): self._chk(lib().orc_finish(self._p))

    def sketch(self):
        mins = np.zeros(self.S, dtype=np.uint64); w = np.zeros(self.S)
        lib().orc_get_sketch(self._p, mins.ctypes.data, w.ctypes.data)
        return mins, w

    def histogram(self):
        h = np.zeros(self.B)
        lib().orc_get_histogram(self._p, h.ctypes.data)
        return h

    def cms(self):
        d, w = cms_geometry()
        a = np.zeros(d * w)
        lib().orc_get_cms(self._p, a.ctypes.data)
        return a.reshape(d, w)

    def used_bins(self): return lib().orc_used_bins(self._p)

    def counters(self):
        v = [ctypes.c_uint64() for _ in range(5)]
        lib().orc_get_counters(self._p, *[ctypes.byref(x) for x in v])
        return dict(zip(("n_reads", "n_minimizers", "total_len", "n_flushes", "n_elements"),
                        [x.value for x in v]))

    def cws(self):
        n = self.S * self.B
        return tuple(np.ctypeslib.as_array(getattr(lib(), f)(self._p), shape=(n,)).reshape(self.S, self.B).copy()
                     for f in ("orc_cws_r", "orc_cws_c", "orc_cws_b"))


def smash_matrix(mins, weights, metric="jaccard"):
    """Pairwise distance matrix out[subject, query] (reference: cmd/smash.go:208-224)."""
    mins = np.ascontiguousarray(mins, dtype=np.uint64); weights = np.ascontiguousarray(weights, dtype=np.float64)
    N, S = mins.shape
    out = np.zeros((N, N))
    lib().orc_smash_matrix(mins.ctypes.data, weights.ctypes.data, N, S, 1 if metric == "weightedjaccard" else 0, out.ctypes.data)
    return out
