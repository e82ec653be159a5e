"""CPU-side checks: the C-ABI library loads and exports every symbol include/hulk_hip.h declares
(no compute calls — there is no GPU here), host-side JSON writer/loader, synthetic reads."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols(experiments=False):
    """entry points include/hulk_hip.h declares: for the shipping library, or (experiments=True) the test hooks it declares
    under #ifdef HULK_EXPERIMENTS — exported by the profiling build only"""
    txt = open(os.path.join(ROOT, "include", "hulk_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    blocks = re.findall(r"#ifdef HULK_EXPERIMENTS\n(.*?)#endif", txt, flags=re.S)
    if experiments:
        txt = "\n".join(blocks)
    else:
        txt = re.sub(r"#ifdef HULK_EXPERIMENTS\n.*?#endif", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hulk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from hulk_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libhulkhip.so first (make -C hulk_amd/csrc)"
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/hulk_hip.h but not exported"
    assert sorted(_lib.ABI_SYMBOLS) == syms, "hulk_amd/_lib.py binding list out of sync with the header"
    L.hulk_abi_version.restype = ctypes.c_int
    assert L.hulk_abi_version() == _lib.HULK_ABI_VERSION == 4
    # the test hooks are not in the shipping library; the profiling build has them and says what it is
    hooks = header_symbols(experiments=True)
    assert hooks == sorted(_lib.EXPERIMENT_SYMBOLS) == ["hulk_debug_inject", "hulk_debug_read"]
    assert not any(hasattr(L, h) for h in hooks)
    L.hulk_build_info.restype = ctypes.c_char_p
    assert b"experiments" not in L.hulk_build_info()
    if os.path.exists(_lib.EXP_LIB_PATH):
        X = ctypes.CDLL(_lib.EXP_LIB_PATH)
        X.hulk_build_info.restype = ctypes.c_char_p
        assert all(hasattr(X, h) for h in hooks + syms) and X.hulk_build_info().endswith(b" experiments=1")
    L.hulk_strerror.restype = ctypes.c_char_p
    assert L.hulk_strerror(-4) == b"sequence length must be >= w + k - 1"
    assert L.hulk_strerror(-5) == b"not used yet"


def test_params_struct_layout_matches_header():
    from hulk_amd._lib import HulkParams
    assert ctypes.sizeof(HulkParams) == 64
    assert HulkParams.decay_ratio.offset == 16 and HulkParams.interval.offset == 24
    assert HulkParams.cws_source.offset == 40 and HulkParams.flags.offset == 44
    hdr = open(os.path.join(ROOT, "include", "hulk_hip.h")).read()
    from hulk_amd import _lib
    for name in ("HULK_FLAG_GAMMA_CPYTHON", "HULK_FLAG_NO_PRUNE", "HULK_FLAG_NO_SKIP", "HULK_FLAG_SHARD_FULL", "HULK_FLAG_NO_OVERLAP",
                 "HULK_FLAG_NO_PRERESERVE", "HULK_FLAG_CMS_CHAIN"):
        assert int(re.search(rf"#define {name} (\d+)u", hdr).group(1)) == getattr(_lib, name)
    assert "#define HULK_MAX_BINS (1 << 20)" in hdr and _lib.HULK_MAX_BINS == 1 << 20


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a gfx950 GPU hulk_create must return an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import hulk_amd
    with pytest.raises(hulk_amd.HulkError):
        hulk_amd.GpuSketcher(21, 9, 8)


def test_parameter_errors_before_device_probe():
    import hulk_amd
    with pytest.raises(hulk_amd.HulkError, match="histosketching only supports k <= 31"):
        hulk_amd.GpuSketcher(32, 9, 8)
    with pytest.raises(hulk_amd.HulkError, match="decay ratio must be between 0.0 and 1.0"):
        hulk_amd.GpuSketcher(21, 9, 8, decay_ratio=1.5)
    with pytest.raises(hulk_amd.HulkError, match="histogram must have at least 2 bins"):
        hulk_amd.GpuSketcher(1, 9, 8)
    with pytest.raises(hulk_amd.HulkError, match="negative value used for number of k-mer spectrum bins"):
        hulk_amd.GpuSketcher(21, 9, 8, num_bins=-3)
    # more bins than the binning kernels' (slot << 20 | bin) keys can hold: refused, never aliased
    with pytest.raises(hulk_amd.HulkError, match="HULK_MAX_BINS"):
        hulk_amd.GpuSketcher(21, 9, 8, num_bins=(1 << 20) + 1)
    with pytest.raises(hulk_amd.HulkError, match="unknown flags"):
        hulk_amd.GpuSketcher(21, 9, 8, flags=1 << 9)


def test_go_float_formatting():
    from hulk_amd.sketchio import go_float
    cases = [(1.7976931348623157e308, "1.7976931348623157e+308"), (0.5, "0.5"), (1e-7, "1e-7"),
             (1.5e-7, "1.5e-7"), (1e21, "1e+21"), (123456789.0, "123456789"), (1e-6, "0.000001"),
             (-0.000001234, "-0.000001234"), (9.999e20, "999900000000000000000"), (100.0, "100"),
             (-3.25e-12, "-3.25e-12"), (0.0, "0"), (-0.1234567890123, "-0.1234567890123"),
             (2.5e-10, "2.5e-10"), (1.2e22, "1.2e+22")]
    for v, want in cases:
        assert go_float(v) == want, (v, go_float(v), want)
    rng = np.random.default_rng(1)
    for v in np.concatenate([rng.standard_normal(200) * 10.0 ** rng.integers(-12, 12, 200)]):
        assert float(go_float(v)) == v            # shortest round-trip digits


def test_json_writer_layout_and_loader(tmp_path):
    from hulk_amd import HistoSketch
    from hulk_amd.sketchio import HULKdata, load_hulk_data, md5sum
    mins = np.array([5, 0, 194480], dtype=np.uint64)
    w = np.array([-0.25, 1.7976931348623157e308, 3e-9])
    hs = HistoSketch(21, mins, w, 194481, False)
    d = HULKdata(); d.add(hs); d.filename = "a.fq,b<c>.fq,"; d.banner_label = "blank"
    txt = d.dumps()
    assert txt.splitlines()[:6] == ['{', '    "class": "hulk_sketch",', '    "filename": "a.fq,b\\u003cc\\u003e.fq,",',
                                    '    "hash_function": "ntHash",', '    "license": "CC0",', '    "signatures": [']
    assert '                "mins": [\n                    5,\n                    0,\n                    194480\n                ],' in txt
    assert '                    1.7976931348623157e+308,\n                    3e-9\n' in txt
    obj = json.loads(txt)
    assert list(obj) == ["class", "filename", "hash_function", "license", "signatures", "version", "banner_label"]
    sk = obj["signatures"][0]
    assert list(sk) == ["Algorithm", "Sketch"]
    assert list(sk["Sketch"]) == ["ksize", "md5sum", "mins", "weights", "num", "num_histogram_bins", "concept_drift"]
    assert sk["Sketch"]["md5sum"] == hashlib.md5(mins.astype("<u8").tobytes()).hexdigest() == md5sum(mins)
    p = tmp_path / "x.json"; d.write_json(p)
    back = load_hulk_data(p)
    assert np.array_equal(back.signatures[0][1].mins, mins)
    bad = txt.replace('"md5sum": "' + sk["Sketch"]["md5sum"], '"md5sum": "' + "0" * 32)
    p.write_text(bad)
    with pytest.raises(ValueError, match="md5sum mismatch"):
        load_hulk_data(p)


def test_synthetic_reads_are_shardable():
    from hulk_amd import synth
    b, o = synth.reads_numpy(0, 40, 150)
    assert set(np.unique(b).tolist()) <= set(b"ACGT") and len(b) == 6000 and o[-1] == 6000
    b2, _ = synth.reads_numpy(17, 5, 150)
    assert np.array_equal(b[17 * 150:22 * 150], b2)
    assert hashlib.sha256(bytes(b[:300])).hexdigest() == hashlib.sha256(bytes(synth.reads_numpy(0, 2, 150)[0])).hexdigest()


def test_spectrum_size_is_int32_pow():
    from hulk_amd import spectrum_size
    assert spectrum_size(21) == 194481 and spectrum_size(31) == 923521
    assert spectrum_size(216) < 0          # int32 wrap, as int32(helpers.Pow(k,4))


def test_khf_signature_layout_and_kmv_is_fatal(tmp_path):
    """`hulk sketch --khf / --kmv` observable behaviour (cmd/sketch.go:58-59, pipeline/boss.go:70-71,
    pipeline/sketch.go:226-234, sketchio.go:56-75): the secondary MinHash sketches are constructed but never fed, so
    KHF is a signature of math.MaxUint64 values (fields ksize, md5sum, mins, num) after the histosketch, and KMV — an
    empty sketch — makes HULKdata.Add fail with the reference's message."""
    from hulk_amd import HistoSketch
    from hulk_amd.sketchio import HULKdata, KHFSketch, KMVSketch, load_hulk_data, md5sum
    d = HULKdata()
    d.add(HistoSketch(21, np.array([3, 9], dtype=np.uint64), np.array([-0.5, -0.25]), 194481, False))
    d.add(KHFSketch(21, 2))
    d.filename, d.banner_label = "x.fq,", "blank"
    obj = json.loads(d.dumps())
    assert [s["Algorithm"] for s in obj["signatures"]] == ["histosketch", "khf"]
    khf = obj["signatures"][1]["Sketch"]
    assert list(khf) == ["ksize", "md5sum", "mins", "num"]
    assert khf["mins"] == [18446744073709551615] * 2 and khf["num"] == 2 and khf["ksize"] == 21
    assert khf["md5sum"] == hashlib.md5(b"\xff" * 16).hexdigest() == md5sum(np.full(2, 2 ** 64 - 1, dtype=np.uint64))
    assert '                "num": 2\n            }\n        }\n    ],' in d.dumps()
    p = tmp_path / "k.json"; d.write_json(p)
    back = load_hulk_data(p)                                       # LoadHULKdata accepts khf signatures (sketchio.go:154-157)
    assert [a for a, _ in back.signatures] == ["histosketch", "khf"]
    with pytest.raises(ValueError, match="no sketch was generated by the kmv algorithm"):
        d.add(KMVSketch(21, 2))


def test_library_is_built_from_this_tree():
    """hulk_build_info(): the .so in the tree reports the hash of the sources in the tree (a stale build — sources edited, `make`
    not run — fails here, on CPU, before any GPU test trusts it) and the compiler it came from."""
    from hulk_amd import _lib
    L = _lib.load()
    info = L.hulk_build_info().decode()
    assert "abi=%d " % _lib.HULK_ABI_VERSION in info and "arch=gfx950" in info and "hipcc=" in info
    assert "sources=" + _lib.source_hash() in info, (info, _lib.source_hash())


def test_cgo_binding_file_is_the_block_of_integration_md():
    """tools/go/gpusketch/gpusketch.go (reviewable source, no Go toolchain here) = INTEGRATION.md's first Go block, and every
    C.hulk_* it calls is an entry point the header declares."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    body = re.search(r"```go\n(.*?)\n```", doc, re.S).group(1)
    src = open(os.path.join(root, "tools", "go", "gpusketch", "gpusketch.go")).read()
    assert src.endswith(body + "\n")
    header = open(os.path.join(root, "include", "hulk_hip.h")).read()
    called = set(re.findall(r"C\.(hulk_[a-z_]+)\(", body))
    assert called and all(re.search(r"\b%s\s*\(" % name, header) for name in called), called


def test_header_is_c99_and_a_c_host_links(tmp_path):
    """cgo compiles the preamble as C: include/hulk_hip.h must be valid C99 (no C++-isms), and a plain C host must link
    against libhulkhip.so and reach the entry points that need no GPU (ABI version, error strings, build info)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "hulk_amd", "csrc")
    src = tmp_path / "host.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "hulk_hip.h"\n'
                   "int main(int argc, char **argv) {\n"
                   "    hulk_params p; memset(&p, 0, sizeof p);\n"
                   "    if (hulk_abi_version() != HULK_ABI_VERSION) return 2;\n"
                   "    if (!strstr(hulk_strerror(HULK_ERR_SHORT_SEQ), \"w + k - 1\")) return 3;\n"
                   '    printf("%s\\n", hulk_build_info());\n'
                   "    /* the sketch-file loader of `hulk smash` needs no GPU: a C host can call it (argv: two sketch files) */\n"
                   "    if (argc == 3) {\n"
                   "        const char *paths[2]; hulk_sketch_set *set = NULL; char err[512]; uint32_t n = 0, size = 0;\n"
                   "        paths[0] = argv[1]; paths[1] = argv[2];\n"
                   "        if (hulk_load_sketches(paths, 2, 21, \"histosketch\", 2, &set, err, sizeof err) != HULK_OK) { printf(\"load: %s\\n\", err); return 4; }\n"
                   "        if (hulk_sketch_set_info(set, &n, &size) != HULK_OK || n != 2 || size != 3) return 5;\n"
                   "        if (hulk_sketch_set_mins(set)[4] != 12u || hulk_sketch_set_weights(set)[5] != -0.5) return 6;\n"
                   "        if (!strstr(hulk_sketch_set_path(set, 1), \"b.json\") || strcmp(hulk_sketch_set_banner(set, 0), \"blank\")) return 7;\n"
                   "        hulk_sketch_set_free(set);\n"
                   "        paths[1] = \"/nonexistent.json\";\n"
                   "        if (hulk_load_sketches(paths, 2, 21, \"histosketch\", 0, &set, err, sizeof err) != HULK_ERR_ARG || set != NULL) return 8;\n"
                   '        printf("loader ok\\n");\n'
                   "    }\n"
                   "    return 0;\n}\n")
    exe = tmp_path / "host"
    subprocess.run(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                    str(src), "-o", str(exe), "-L", libdir, "-lhulkhip", "-Wl,-rpath," + libdir], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.returncode, p.stderr)
    assert "abi=" in p.stdout and "arch=gfx950" in p.stdout
    from hulk_amd.sketchio import HULKdata, HistoSketch
    files = []
    for name, mins, wts in (("a.json", [3, 7, 11], [1.5, -2.0, 1e-9]), ("b.json", [5, 12, 0], [0.25, 4.0, -0.5])):
        d = HULKdata(); d.filename = name; d.banner_label = "blank"
        d.add(HistoSketch(21, np.array(mins, dtype=np.uint64), np.array(wts), 21 ** 4, False))
        d.write_json(tmp_path / name)
        files.append(str(tmp_path / name))
    p = subprocess.run([str(exe)] + files, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "loader ok" in p.stdout, (p.returncode, p.stdout, p.stderr)


def test_rccl_test_double_exports_what_the_library_binds():
    """tests/cpp/libfakerccl.so (built by __graft_entry__.build(); the stand-in for librccl.so.1 of tests/test_gpu_fake_rccl.py)
    exports every nccl* symbol hulk_comm.hip resolves with dlsym — read from the source, so a new binding cannot be forgotten."""
    lib = os.path.join(ROOT, "tests", "cpp", "libfakerccl.so")
    if not os.path.exists(lib):
        pytest.skip("libfakerccl.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    src = open(os.path.join(ROOT, "hulk_amd", "csrc", "hulk_comm.hip")).read()
    bound = set(re.findall(r"RCCL_SYM\((\w+)\)", src)) - {"f"}
    assert len(bound) == 8, bound
    F = ctypes.CDLL(lib)
    for name in bound:
        assert hasattr(F, "nccl" + name), name
    assert hasattr(F, "fakeRcclStats")
