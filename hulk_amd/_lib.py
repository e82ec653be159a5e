"""ctypes binding of libhulkhip.so (the C ABI declared in include/hulk_hip.h).

The library is built in-tree (hulk_amd/csrc/Makefile).  There is NO fallback: if the shared
object is missing or no gfx950 GPU is usable, importing/creating fails loudly.

One HIP runtime per process: torch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's, which
libhulkhip.so is linked against).  Whichever is loaded first serves both, and torch on top of /opt/rocm's runtime finds
no device — so when torch is installed, load() maps torch's copy first (without importing torch) and libhulkhip binds
to it; `import torch` may then come before or after the first hulk_amd call.  Without torch /opt/rocm's is used.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhulkhip.so")
# the profiling build (make -C hulk_amd/csrc EXPERIMENTS=1: the experiment switches of docs/EXPERIMENTS.md compiled in);
# tools/ select it with HULK_LIB=exp, or HULK_LIB=<path> for any other build of the library
EXP_LIB_PATH = os.path.join(_HERE, "csrc", "libhulkhip_exp.so")
if os.environ.get("HULK_LIB"):
    LIB_PATH = EXP_LIB_PATH if os.environ["HULK_LIB"] == "exp" else os.environ["HULK_LIB"]

HULK_OK = 0
HULK_ABI_VERSION = 4            # include/hulk_hip.h; load() refuses a libhulkhip.so built from another version of the header
HULK_UNIQUE_ID_BYTES = 128
HULK_XCHG_ALLGATHER, HULK_XCHG_ALLREDUCE_U32 = 0, 1
HULK_CWS_GO_COMPAT = 0
HULK_CWS_EXTERNAL = 1
HULK_FLAG_GAMMA_CPYTHON, HULK_FLAG_NO_PRUNE, HULK_FLAG_NO_SKIP, HULK_FLAG_SHARD_FULL, HULK_FLAG_NO_OVERLAP, HULK_FLAG_NO_PRERESERVE, HULK_FLAG_CMS_CHAIN = 1, 2, 4, 8, 16, 32, 64
HULK_MAX_BINS = 1 << 20
HULK_INJECT_NONE, HULK_INJECT_STALE_SEAL, HULK_INJECT_STALE_STAGE = 0, 1, 2
HULK_DEBUG_TILEMIN, HULK_DEBUG_SCANMAP = 1, 2

# every symbol include/hulk_hip.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = (
    "hulk_abi_version", "hulk_build_info", "hulk_strerror", "hulk_last_error", "hulk_create", "hulk_destroy",
    "hulk_set_stream", "hulk_set_private_stream", "hulk_set_cws_tables", "hulk_add_reads", "hulk_add_reads_device",
    "hulk_batch_size", "hulk_bin_reads_device", "hulk_bin_reads_device_at", "hulk_histogram_device", "hulk_flush_batch", "hulk_flush_batch_after", "hulk_add_histogram", "hulk_flush",
    "hulk_finish", "hulk_get_sketch", "hulk_get_counters", "hulk_get_histogram", "hulk_get_cms",
    "hulk_get_cws_tables", "hulk_smash", "hulk_smash_ex", "hulk_selftest_reciprocal", "hulk_set_profiling", "hulk_get_profile",
    "hulk_parse_files", "hulk_sketch_files", "hulk_parse_files_opts", "hulk_sketch_files_opts", "hulk_get_scan_stats", "hulk_synchronize",
    "hulk_comm_unique_id", "hulk_comm_init", "hulk_comm_init_host", "hulk_comm_init_loopback", "hulk_step_sharded", "hulk_step_sharded_host",
    "hulk_step_sliced", "hulk_gather_sketch", "hulk_get_comm_stats", "hulk_get_comm_health", "hulk_release_caches", "hulk_get_device_checks", "hulk_get_profile_table",
    "hulk_load_sketches", "hulk_sketch_set_free", "hulk_sketch_set_info", "hulk_sketch_set_mins", "hulk_sketch_set_weights", "hulk_sketch_set_path",
    "hulk_sketch_set_banner", "hulk_smash_files",
)
# test hooks: exported by the profiling build only (make -C hulk_amd/csrc EXPERIMENTS=1; HULK_LIB=exp)
EXPERIMENT_SYMBOLS = ("hulk_debug_inject", "hulk_debug_read")


class HulkParams(ctypes.Structure):
    _fields_ = [
        ("k", ctypes.c_uint32), ("w", ctypes.c_uint32), ("sketch_size", ctypes.c_uint32),
        ("num_bins", ctypes.c_int32), ("decay_ratio", ctypes.c_double),
        ("interval", ctypes.c_uint32), ("device", ctypes.c_int32),
        ("slot_begin", ctypes.c_uint32), ("slot_count", ctypes.c_uint32),
        ("cws_source", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("batch", ctypes.c_uint32), ("work_lanes", ctypes.c_uint32),
        ("host_copy_threads", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
    ]


class IngestStats(ctypes.Structure):
    _fields_ = [("n_seqs", ctypes.c_uint64), ("total_len", ctypes.c_uint64), ("n_lines", ctypes.c_uint64),
                ("bytes_in", ctypes.c_uint64), ("seconds", ctypes.c_double)]


HULK_INGEST_GZ_ONE_THREAD, HULK_INGEST_GZ_ZLIB, HULK_INGEST_TRACE, HULK_INGEST_HOST_PARSER = 1, 2, 4, 8


class IngestOpts(ctypes.Structure):
    """hulk_ingest_opts (include/hulk_hip.h): the knobs of one run of the host ingest, 0 = default."""
    _fields_ = [("parser_threads", ctypes.c_uint32), ("gz_threads", ctypes.c_uint32), ("file_readers", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("block_bytes", ctypes.c_uint64), ("gz_chunk_bytes", ctypes.c_uint64),
                ("reserved", ctypes.c_uint64 * 2)]


BATCH_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8),
                            ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64)
# hulk_exchange_fn: (user, op, send, recv, bytes) -> 0 on success
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)


class SmashStats(ctypes.Structure):
    _fields_ = [("seconds_load", ctypes.c_double), ("seconds_matrix", ctypes.c_double), ("seconds_csv", ctypes.c_double),
                ("kernel_ms", ctypes.c_double), ("n_sketches", ctypes.c_uint32), ("sketch_size", ctypes.c_uint32)]


class HulkError(RuntimeError):
    """Carries the message the reference would have passed to log.Fatalf("ERROR---> %v")."""
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code
        self.message = message


_lib = None


def _soname(path):
    """DT_SONAME of an ELF shared object (None if it cannot be read)."""
    import subprocess
    try:
        out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, timeout=20).stdout
    except (OSError, subprocess.SubprocessError):
        return None
    for line in out.splitlines():
        if "(SONAME)" in line and "[" in line:
            return line.split("[", 1)[1].split("]", 1)[0]
    return None


def _needed(path, prefix):
    import subprocess
    try:
        out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, timeout=20).stdout
    except (OSError, subprocess.SubprocessError):
        return None
    for line in out.splitlines():
        if "(NEEDED)" in line and "[" + prefix in line:
            return line.split("[", 1)[1].split("]", 1)[0]
    return None


HIP_RUNTIME_BOUND = None        # which HIP runtime load() mapped first: "torch:<path>", "system" or "already loaded (torch)"


def _preload_torch_hip_runtime():
    """Map torch's bundled HIP runtime before libhulkhip.so pulls in /opt/rocm's (see the module docstring).
    HULK_NO_TORCH_PRELOAD=1 opts out; a bundled runtime whose SONAME is not the one libhulkhip.so was linked against
    is left alone (both would end up loaded, and the second one sees no device) with a warning."""
    global HIP_RUNTIME_BOUND
    import importlib.util
    import sys
    import warnings
    HIP_RUNTIME_BOUND = "system"
    if os.environ.get("HULK_NO_TORCH_PRELOAD"):
        return
    if "torch" in sys.modules:
        HIP_RUNTIME_BOUND = "already loaded (torch)"
        return                                   # torch is loaded: its runtime already owns the SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        want, have = _needed(LIB_PATH, "libamdhip64"), _soname(path)
        if want and have and want != have:
            warnings.warn(f"hulk_amd: torch bundles HIP runtime {have} but libhulkhip.so is linked against {want}; not "
                          "preloading it — import torch AFTER the first hulk_amd call will find no device "
                          "(rebuild libhulkhip.so against torch's ROCm, or set HULK_NO_TORCH_PRELOAD=1 to silence)")
            return
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            HIP_RUNTIME_BOUND = "torch:" + path
        except OSError as e:
            raise ImportError(f"torch is installed but its HIP runtime {path} does not load ({e}); importing torch "
                              "before hulk_amd is the work-around") from e


def source_hash():
    """First 16 hex digits of the SHA-256 over hulk_amd/csrc's sources and headers, in the Makefile's order — what
    hulk_build_info() of a library built from this tree reports."""
    import hashlib
    import re
    csrc = os.path.join(_HERE, "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    names = re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split() + re.search(r"^HDRS = (.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(csrc, n), "rb").read())
    return h.hexdigest()[:16]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C hulk_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). hulk_amd has no CPU fallback.")
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, i32, dbl = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_double
    L.hulk_abi_version.restype = ctypes.c_int
    if L.hulk_abi_version() != HULK_ABI_VERSION:
        raise ImportError(f"{LIB_PATH} reports ABI version {L.hulk_abi_version()}, this binding is written for "
                          f"{HULK_ABI_VERSION}: rebuild it (`make -C hulk_amd/csrc`)")
    L.hulk_build_info.restype = ctypes.c_char_p
    L.hulk_strerror.restype = ctypes.c_char_p; L.hulk_strerror.argtypes = [ctypes.c_int]
    L.hulk_last_error.restype = ctypes.c_char_p; L.hulk_last_error.argtypes = [vp]
    L.hulk_create.restype = ctypes.c_int; L.hulk_create.argtypes = [ctypes.POINTER(HulkParams), ctypes.POINTER(vp)]
    L.hulk_destroy.restype = None; L.hulk_destroy.argtypes = [vp]
    L.hulk_set_stream.restype = ctypes.c_int; L.hulk_set_stream.argtypes = [vp, vp]
    L.hulk_set_private_stream.restype = ctypes.c_int; L.hulk_set_private_stream.argtypes = [vp]
    L.hulk_set_cws_tables.restype = ctypes.c_int; L.hulk_set_cws_tables.argtypes = [vp, vp, vp, vp]
    L.hulk_add_reads.restype = ctypes.c_int; L.hulk_add_reads.argtypes = [vp, vp, vp, u64]
    L.hulk_add_reads_device.restype = ctypes.c_int; L.hulk_add_reads_device.argtypes = [vp, vp, vp, u64, u32, u64]
    L.hulk_bin_reads_device.restype = ctypes.c_int; L.hulk_bin_reads_device.argtypes = [vp, vp, vp, u64, u32, u64, u64]
    L.hulk_bin_reads_device_at.restype = ctypes.c_int; L.hulk_bin_reads_device_at.argtypes = [vp, vp, vp, u64, u32, u64, u64, u32]
    L.hulk_batch_size.restype = u32; L.hulk_batch_size.argtypes = [vp]
    L.hulk_flush_batch.restype = ctypes.c_int; L.hulk_flush_batch.argtypes = [vp, u32]
    L.hulk_flush_batch_after.restype = ctypes.c_int; L.hulk_flush_batch_after.argtypes = [vp, u32, vp]
    L.hulk_histogram_device.restype = vp; L.hulk_histogram_device.argtypes = [vp]
    L.hulk_add_histogram.restype = ctypes.c_int; L.hulk_add_histogram.argtypes = [vp, vp]
    L.hulk_flush.restype = ctypes.c_int; L.hulk_flush.argtypes = [vp]
    L.hulk_finish.restype = ctypes.c_int; L.hulk_finish.argtypes = [vp]
    L.hulk_synchronize.restype = ctypes.c_int; L.hulk_synchronize.argtypes = [vp]
    L.hulk_get_sketch.restype = ctypes.c_int; L.hulk_get_sketch.argtypes = [vp, vp, vp]
    L.hulk_get_counters.restype = ctypes.c_int; L.hulk_get_counters.argtypes = [vp, vp, vp, vp]
    L.hulk_get_histogram.restype = ctypes.c_int; L.hulk_get_histogram.argtypes = [vp, vp]
    L.hulk_get_cms.restype = ctypes.c_int; L.hulk_get_cms.argtypes = [vp, vp]
    L.hulk_get_cws_tables.restype = ctypes.c_int; L.hulk_get_cws_tables.argtypes = [vp, vp, vp, vp]
    L.hulk_smash.restype = ctypes.c_int; L.hulk_smash.argtypes = [ctypes.c_int, vp, vp, u32, u32, ctypes.c_int, vp]
    L.hulk_smash_ex.restype = ctypes.c_int; L.hulk_smash_ex.argtypes = [ctypes.c_int, vp, vp, u32, u32, ctypes.c_int, vp, vp]
    L.hulk_selftest_reciprocal.restype = ctypes.c_int; L.hulk_selftest_reciprocal.argtypes = [vp, vp]
    L.hulk_set_profiling.restype = ctypes.c_int; L.hulk_set_profiling.argtypes = [vp, ctypes.c_int]
    L.hulk_get_profile.restype = ctypes.c_int; L.hulk_get_profile.argtypes = [vp, ctypes.c_char_p, vp, vp]
    L.hulk_get_scan_stats.restype = ctypes.c_int; L.hulk_get_scan_stats.argtypes = [vp, vp, vp]
    L.hulk_parse_files.restype = ctypes.c_int
    L.hulk_parse_files.argtypes = [ctypes.POINTER(ctypes.c_char_p), u32, ctypes.c_int, u32, BATCH_FN, vp,
                                   ctypes.POINTER(IngestStats), ctypes.c_char_p, u64]
    L.hulk_sketch_files.restype = ctypes.c_int
    L.hulk_sketch_files.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), u32, ctypes.c_int, u32,
                                    ctypes.POINTER(IngestStats)]
    L.hulk_parse_files_opts.restype = ctypes.c_int
    L.hulk_parse_files_opts.argtypes = [ctypes.POINTER(ctypes.c_char_p), u32, ctypes.c_int, ctypes.POINTER(IngestOpts), BATCH_FN, vp,
                                        ctypes.POINTER(IngestStats), ctypes.c_char_p, u64]
    L.hulk_sketch_files_opts.restype = ctypes.c_int
    L.hulk_sketch_files_opts.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), u32, ctypes.c_int, ctypes.POINTER(IngestOpts),
                                         ctypes.POINTER(IngestStats)]
    L.hulk_comm_unique_id.restype = ctypes.c_int; L.hulk_comm_unique_id.argtypes = [vp]
    L.hulk_comm_init.restype = ctypes.c_int; L.hulk_comm_init.argtypes = [vp, vp, u32, u32]
    L.hulk_comm_init_host.restype = ctypes.c_int; L.hulk_comm_init_host.argtypes = [vp, u32, u32, EXCHANGE_FN, vp]
    L.hulk_comm_init_loopback.restype = ctypes.c_int; L.hulk_comm_init_loopback.argtypes = [vp, u32, u32]
    L.hulk_step_sharded.restype = ctypes.c_int; L.hulk_step_sharded.argtypes = [vp, vp, vp, u64, u32, u64, u32]
    L.hulk_step_sharded_host.restype = ctypes.c_int; L.hulk_step_sharded_host.argtypes = [vp, vp, vp, u64, u32]
    L.hulk_step_sliced.restype = ctypes.c_int; L.hulk_step_sliced.argtypes = [vp, vp, vp, u64, u32, u64, u64, u32]
    L.hulk_gather_sketch.restype = ctypes.c_int; L.hulk_gather_sketch.argtypes = [vp, vp, vp]
    L.hulk_get_comm_stats.restype = ctypes.c_int; L.hulk_get_comm_stats.argtypes = [vp, vp, vp, vp]
    L.hulk_get_comm_health.restype = ctypes.c_int; L.hulk_get_comm_health.argtypes = [vp, vp, vp]
    if hasattr(L, "hulk_debug_inject"):                 # the profiling build (hulk_build_info ends in " experiments=1")
        L.hulk_debug_inject.restype = ctypes.c_int; L.hulk_debug_inject.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint64]
        L.hulk_debug_read.restype = ctypes.c_int; L.hulk_debug_read.argtypes = [vp, ctypes.c_uint32, vp, vp]
    cpp = ctypes.POINTER(ctypes.c_char_p)
    L.hulk_load_sketches.restype = ctypes.c_int
    L.hulk_load_sketches.argtypes = [cpp, u32, u32, ctypes.c_char_p, u32, ctypes.POINTER(vp), ctypes.c_char_p, u64]
    L.hulk_sketch_set_free.restype = None; L.hulk_sketch_set_free.argtypes = [vp]
    L.hulk_sketch_set_info.restype = ctypes.c_int; L.hulk_sketch_set_info.argtypes = [vp, vp, vp]
    L.hulk_sketch_set_mins.restype = vp; L.hulk_sketch_set_mins.argtypes = [vp]
    L.hulk_sketch_set_weights.restype = vp; L.hulk_sketch_set_weights.argtypes = [vp]
    L.hulk_sketch_set_path.restype = ctypes.c_char_p; L.hulk_sketch_set_path.argtypes = [vp, u32]
    L.hulk_sketch_set_banner.restype = ctypes.c_char_p; L.hulk_sketch_set_banner.argtypes = [vp, u32]
    L.hulk_smash_files.restype = ctypes.c_int
    L.hulk_smash_files.argtypes = [ctypes.c_int, cpp, u32, u32, ctypes.c_char_p, ctypes.c_char_p, u32, ctypes.c_char_p, ctypes.c_char_p, vp,
                                   ctypes.POINTER(SmashStats), ctypes.c_char_p, u64]
    L.hulk_get_device_checks.restype = ctypes.c_int; L.hulk_get_device_checks.argtypes = [vp, vp, vp]
    L.hulk_get_profile_table.restype = ctypes.c_int; L.hulk_get_profile_table.argtypes = [vp, ctypes.c_char_p, u64]
    L.hulk_release_caches.restype = ctypes.c_int; L.hulk_release_caches.argtypes = []
    _lib = L
    return L


def is_experiments_build() -> bool:
    """True when the loaded library is the profiling build (libhulkhip_exp.so: experiment switches and test hooks compiled in)."""
    return b"experiments=1" in load().hulk_build_info()
