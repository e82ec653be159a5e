import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, hulk_amd
from hulk_amd import synth
d = tempfile.mkdtemp(dir="/dev/shm")
path = os.path.join(d, "c.fa")
with open(path, "wb") as fh:
    for i in range(200):
        seq = synth.reads_numpy(i, 1, 500000)[0].tobytes()
        fh.write(b">c%d\n" % i + b"\n".join(seq[j:j+60] for j in range(0, 500000, 60)) + b"\n")
for rep in range(3):
    sk = hulk_amd.GpuSketcher(21, 9, 512, interval=0)
    t0 = time.perf_counter(); st = sk.sketch_files([path], fasta=True); t1 = time.perf_counter(); sk.finish(); t2 = time.perf_counter()
    print("run", rep, "sketch_files %.1f ms, finish %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3), st["seconds"])
    sk.close()
from hulk_amd import ingest
for th in (1, 4, 16):
    t0 = time.perf_counter(); b, o, st = ingest.parse_files([path], fasta=True, threads=th, collect=False); print("parse only threads", th, "%.1f ms" % ((time.perf_counter()-t0)*1e3))
import shutil; shutil.rmtree(d)
