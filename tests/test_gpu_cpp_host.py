"""The C++ host mirror (include/hulk.hpp: Boss / HistoSketch with the reference's method names) built
with g++ against libhulkhip.so and compared with the ctypes path and the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, pack_reads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "hulk_amd", "csrc")


def build(tmp_path):
    exe = str(tmp_path / "boss_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "boss_driver.cpp"), "-o", exe,
           "-L", LIBDIR, "-lhulkhip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def test_cpp_host_header_compiles_and_links(tmp_path):
    """CPU side: hulk.hpp is valid C++17 and every symbol it uses resolves against libhulkhip.so."""
    build(tmp_path)


@pytest.mark.gpu
def test_cpp_boss_matches_python_and_oracle(tmp_path, fq_reads):
    import hulk_amd
    from oracle import pyorc
    exe = build(tmp_path)
    txt = tmp_path / "reads.txt"
    txt.write_bytes(b"\n".join(fq_reads) + b"\n")
    for k, S, interval, decay in ((21, 64, 0, 1.0), (15, 32, 250, 0.05)):
        g = hulk_amd.GpuSketcher(k, 9, S, interval, decay)
        g.add_reads(*pack_reads(fq_reads))
        g.finish()
        m, w = g.sketch()
        modes = [("addseq", str(txt)), ("files", os.path.join(GOLDEN, "test-reads-small.fq.gz"))]
        if interval:                                    # hulk::Boss::Shard: RCCL communicator (world size 1) from C++,
            modes.append(("sharded", str(txt)))         # hulk_step_sharded_host per full share, ragged last step
        for mode, path in modes:
            p = subprocess.run([exe, mode, path, str(k), "9", str(S), str(interval), repr(decay)],
                               capture_output=True, text=True, timeout=300)
            assert p.returncode == 0, p.stdout + p.stderr
            doc = json.loads(p.stdout.strip().splitlines()[-1])       # (RCCL prints its banner on stdout when it is bound)
            assert doc["n_seqs"] == 1000 and doc["n_minimizers"] == g.get_minimizer_count()
            assert doc["ksize"] == k and doc["num"] == S and doc["bins"] == k ** 4 and doc["drift"] == (decay != 1.0)
            assert np.array_equal(np.array(doc["mins"], dtype=np.uint64), m)
            assert np.array_equal(np.array(doc["weights"]), w)              # %.17g round-trips
        o = pyorc.Sketcher(k, 9, S, 0, decay, interval)
        for r in fq_reads:
            o.add_read(r)
        o.finish()
        assert np.array_equal(o.sketch()[0], m)
        g.close(); o.close()


@pytest.mark.gpu
def test_cpp_errors_carry_the_reference_text(tmp_path):
    exe = build(tmp_path)
    p = subprocess.run([exe, "errors"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines == ["-1|w must be: 0 < w < 257", "-6|histosketching only supports k <= 31",
                     "-7|decay ratio must be between 0.0 and 1.0", "-4|sequence length must be >= w + k - 1"]


@pytest.mark.gpu
def test_cpp_smash_files_writes_the_python_forms_csv(tmp_path):
    """hulk::SmashFiles (include/hulk.hpp over hulk_smash_files): the directory form of `hulk smash` from a C++ host — the CSV is
    byte for byte the one the Python form writes, a corrupted file raises hulk::Error with the reference's text."""
    import hulk_amd
    from hulk_amd import smash as smash_mod, synth
    from hulk_amd.sketchio import HULKdata
    exe = build(tmp_path)
    d = tmp_path / "sk"
    d.mkdir()
    files = []
    for i, first in enumerate((0, 2000, 70000)):
        g = hulk_amd.GpuSketcher(15, 9, 48, interval=1000)
        g.add_reads(*synth.reads_numpy(first, 3000, 120))
        g.finish()
        doc = HULKdata(); doc.add(g.histosketch()); doc.filename = f"r{i}.fq,"; doc.banner_label = "blank"
        p = d / f"s{i}.json"
        doc.write_json(p)
        files.append(str(p))
        g.close()
    out_cpp, out_py = str(tmp_path / "cpp.csv"), str(tmp_path / "py")
    r = subprocess.run([exe, "smashfiles", out_cpp, "weightedjaccard", "15"] + files[::-1], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    order, dist = smash_mod.smash_python(str(d), out_py, 15, "histosketch", "weightedjaccard")
    assert info["n"] == 3 and info["size"] == 48 and info["d01"] == dist[0, 1]
    assert open(out_cpp, "rb").read() == open(out_py + ".hulk-matrix.csv", "rb").read()
    bad = d / "t_bad.json"
    bad.write_text((d / "s0.json").read_text().replace('"version": "1.0.0"', '"version": "2.0.0"'))
    r = subprocess.run([exe, "smashfiles", out_cpp, "jaccard", "15"] + files + [str(bad)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "the loaded sketch was created with a different version of HULK: 2.0.0" in (r.stdout + r.stderr)
