#!/usr/bin/env python3
"""bench.py — reads/sec of the HULK `sketch` hot path on MI355X (BASELINE.json metric).

Workload at N=1 (BASELINE configs[1], "C2"): synthetic 150 bp reads, k=21, w=9, sketchSize=512,
interval=100k reads; reads are resident in HBM before the timed region.  One *step* = one batch of
T=16 sketching intervals per rank (1.6 M reads): one launch chain bins the reads of the 16
intervals into 16 k-mer spectra (minimizers -> jump hash -> spectrum), the spectra go through the
count-min update, and ONE pass over the CWS table applies all 16 histosketch updates in interval order —
bit-identical to flushing after every 100k reads (tests/test_gpu_parity.py).  The flush of step n
runs on a second stream under the minimizer kernels of step n+1.  Default K=20 steps = 32 M reads
(C2's 10 M reads = 6.25 steps; `value_cold` is C2 exactly as stated: 10 M reads, fresh context, no warm-up).

N>1: one process per GPU; the exchange between the ranks is INSIDE libhulkhip.so (hulk_comm_init: RCCL over xGMI;
hulk_step_sharded / hulk_step_sliced, include/hulk_hip.h) — torch.distributed only carries the rendezvous (the 128-byte
RCCL id), the barriers around the timed region and the MAX over the ranks' clocks.  Launched by torch.distributed.run
(RANK/WORLD_SIZE in the environment) this process is one rank; launched as plain `python bench.py --gpus N` it spawns the
N ranks itself and relays their line.  The sketching interval is always 100k reads of the GLOBAL stream (the reference's
rule, pipeline/sketch.go:211-215), so every mode but `sliced-weak` computes the sketch ONE GPU computes over the same stream:
  sharded (headline): a step = N x 16 intervals of the global stream, rank g bins the WHOLE intervals [16g, 16g+16) of it
      (1.6 M reads per rank per step at every N: `scaling` is "weak" in the driver's sense — the stream an N-rank run gets
      through in K steps is N times longer — while the sketch stays the single-GPU one, `sketch_md5` of the same stream);
      count-min is replicated, the CWS update slot-sharded; per step ONE all-gather: of the spectra while an element can
      still change a weight (the first step), of the count-min increments (56 KB per interval) afterwards.
  sliced-strong (SURVEY.md 8e to the letter): a step = 16 intervals, rank g bins reads [g*I/N, (g+1)*I/N) of every interval,
      ONE all-reduce (uint32 sum) of the 16 spectra per step; total work fixed as N grows.
  sliced-weak: every rank bins 100k reads per interval, i.e. the global interval is N x 100k — another sketch than C2's.
  At N > 1 the modes not used for the headline are timed too (`other_scaling`), and `value_c4` is BASELINE configs[3]:
  50 M reads per rank (400 M at N = 8) through the sharded steps on fresh contexts, ragged last step, the clock stopped
  after the EOF gather of the sketch.
HULK_BENCH_TRANSPORT=gloo (test aid): the ranks share GPU 0 and the library's HOST transport carries the exchange over gloo
(RCCL refuses two ranks on one device) — same protocol, same kernels; tests/test_gpu_bench_contract.py runs world 2 this way.
HULK_BENCH_TRANSPORT=fakerccl (test aid): the ranks share GPU 0 and go through the library's RCCL branch (hulk_comm_init:
ncclCommInitRank, grouped ncclAllGather / ncclAllReduce on the flush stream) with the nccl* symbols bound to the test double
tests/cpp/libfakerccl.so (HULK_RCCL_LIB); torch.distributed (control plane: barriers, the unique id) runs over gloo.

Passes, in order: discarded ones (the first pass of a process measures low, and a GPU fresh from idle for seconds: one
pass + HULK_BENCH_PREWARM_S = 2 s of them), then the HEADLINE (W warm-up + K timed steps between barriers, no event
brackets in it), then the secondary legs, each of them fault-isolated: `kernels` (the same steps with every kernel alone on
one stream and bracketed by HIP events: the durations the roofline objects are computed from), `value_unpruned`,
`ms_per_step_long` (>= 200 steps), at N = 1 `value_cold`, `c3`, `c5`, the end-to-end file figures and the CPU baseline, at
N > 1 the other modes and `value_c4`.  A leg that raises leaves `"<leg>_error": "<text>"` in the line and the run goes on;
a leg that hangs (a rank stuck in a collective) is ended by a watchdog on every rank after HULK_BENCH_LEG_TIMEOUT_S
(default 300) seconds: rank 0 prints the line with what it has and every rank exits 0.  HULK_BENCH_FAIL=<leg>[,<leg>] makes
the named legs raise (test aid).  Prints ONE JSON line on rank 0; the exit status is non-zero only if the headline
itself did not run.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, W, S, READ_LEN, INTERVAL = 21, 9, 512, 150, 100_000
NUM_BINS = K ** 4        # cmd/sketch.go:118
STEPTIMES = int(os.environ.get("HULK_BENCH_STEPTIMES", "0"))     # diagnosis: per-step wall times of every pass on stderr
RAMP_MS = float(os.environ.get("HULK_BENCH_RAMP_MS", "40"))   # ms of elementwise kernels on a scratch tensor right before every pass's warm-up
PREWARM_S = float(os.environ.get("HULK_BENCH_PREWARM_S", "2"))   # seconds of discarded passes before the timed ones
BATCH = int(os.environ.get("HULK_BENCH_BATCH", "16"))   # sketching intervals per step (one pass over the CWS table)
C2_READS = 10_000_000        # BASELINE configs[1]
C4_READS_PER_RANK = int(os.environ.get("HULK_BENCH_C4_READS_PER_RANK", "50000000"))   # BASELINE configs[3]: 400 M reads on 8 GPUs
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
SIMDS, CLOCK_GHZ, VALU_CYCLES = 256 * 4, 2.4, 2   # MI355X_MICROARCH.md: 4 SIMD-32 per CU, a wave64 VALU op issues over 2 cycles
def _newest(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return os.path.join("profiles", n)
    return os.path.join("profiles", names[-1])


PMC_PROFILE = _newest("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json")
ISA_MIX = _newest("r04_isa_mix.json", "r03_isa_mix.json")   # tools/isa_mix.py over the production kernels, priced by profiles/r03_op_cost.txt
LEG_TIMEOUT_S = float(os.environ.get("HULK_BENCH_LEG_TIMEOUT_S", "300"))
FAIL_LEGS = set(x for x in os.environ.get("HULK_BENCH_FAIL", "").split(",") if x)
HANG_LEGS = set(x for x in os.environ.get("HULK_BENCH_HANG", "").split(",") if x)      # test aid: the named legs never return
LONG_STEPS = int(os.environ.get("HULK_BENCH_LONG_STEPS", "200"))
C3 = dict(k=31, w=9, S=1024, decay=0.02, interval=100_000,                         # BASELINE configs[2] at its stated 50 M reads
          reads=int(os.environ.get("HULK_BENCH_C3_READS", "50000000")))
C5 = dict(n=1024, S=2048)                                                         # BASELINE configs[4]
CLOCK_MEASURED_GHZ = 2.3    # shader clock under VALU load: s_memtime ticks per ns of HIP-event time, tools/ubench/op_cost2.hip (2.1-2.35)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_intervals=4):
    """The CPU oracle (oracle/hulk_oracle.c, a literal port of the Go algorithm) timed on this box's host cores on a
    bounded sample of the same workload (SURVEY.md §8d "CPU baseline beside it"):
      leg (i)  everything on one thread;
      leg (ii) the minimizer + jump-hash stage of every interval on all host cores (one private k-mer spectrum per
               thread, summed), the count-min + histosketch update single-threaded as in the Go reference (one
               Sketcher goroutine, pipeline/sketch.go:271-301).
    Both legs sketch the first `sample_intervals` intervals of the bench stream on a fresh sketch, so they include the
    one interval (the first) in which most slots still change; per-interval cost does not depend on that (AddElement
    evaluates every slot for every bin either way), so the rate scales linearly to any number of reads."""
    from oracle import pyorc
    from hulk_amd import synth
    n = sample_intervals * INTERVAL
    bases, offsets = synth.reads_numpy(0, n, READ_LEN)
    nproc = os.cpu_count() or 1
    nthreads = min(nproc, 64)       # every host core up to 64 threads: the minimizer stage is < 5 % of the CPU time, and a thread
                                    # per logical core of a 256-thread box only adds start-up cost (32 -> 64 threads: no change)
    # ---- leg (i): one thread
    o = pyorc.Sketcher(K, W, S, 0, 1.0, INTERVAL)          # CWS table generation: not timed (one-off, as in the GPU figure)
    t0 = time.perf_counter()
    o.add_reads(bases, offsets)
    dt1 = time.perf_counter() - t0
    m1, _ = o.sketch()
    # ---- leg (ii): binning on all cores, histosketch on one
    o2 = pyorc.Sketcher(K, W, S, 0, 1.0, 0)
    workers = [pyorc.Sketcher(K, W, 1, 0, 1.0, 0) for _ in range(nthreads)]       # S=1: only their k-mer spectrum is used
    t0 = time.perf_counter()
    for t in range(sample_intervals):
        cuts = np.linspace(t * INTERVAL, (t + 1) * INTERVAL, nthreads + 1).astype(np.int64)
        hists = [None] * nthreads

        def work(i):
            a, b = int(cuts[i]), int(cuts[i + 1])
            if b > a:
                lo = int(offsets[a])
                workers[i].add_reads(bases[lo:int(offsets[b])], offsets[a:b + 1] - offsets[a])   # ctypes releases the GIL
            hists[i] = workers[i].histogram()
            workers[i].wipe()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        o2.add_histogram(np.sum(hists, axis=0).astype(np.uint32))
        o2.flush()
    dt2 = time.perf_counter() - t0
    m2, _ = o2.sketch()
    assert np.array_equal(m1, m2), "all-cores CPU leg disagrees with the single-threaded one"
    o.close(); o2.close()
    for x in workers:
        x.close()
    what = (f"{n} reads = {sample_intervals} intervals of {INTERVAL} (k={K}, sketchSize={S}), fresh sketch (includes the "
            f"first interval), C port of the Go path (oracle/hulk_oracle.c), CWS table generation excluded")
    return {"value": n / dt2, "unit": "reads/s", "cores": nthreads, "kind": "port",
            "sample": what + f"; minimizer + jump-hash stage on {nthreads} threads (box: {nproc} logical cores), count-min + histosketch on 1 (as the "
                             f"reference's single Sketcher goroutine); {dt2:.1f} s",
            "nproc": nproc, "cpu_model": cpu_model(),
            "single_thread": {"value": n / dt1, "unit": "reads/s", "cores": 1, "seconds": dt1},
            "reference_binary": "unavailable: no Go toolchain on this box, the reference's modules are not vendored"}


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) and relay their JSON line."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("HULK_BENCH_TRANSPORT") not in ("gloo", "fakerccl"):      # (test aids: the ranks share GPU 0)
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        sys.stderr.write(p.stdout)
        raise SystemExit(p.returncode or 1)
    print(lines[0], flush=True)
    raise SystemExit(0)


def e2e_file_rates(n_reads=2_000_000):
    """End to end, FASTQ file -> sketch (the reference's DataStreamer/FastqHandler/AddSeq loop, pipeline/sketch.go:40-217,
    in native code: hulk_sketch_files): a synthetic FASTQ of `n_reads` 150 bp reads on /dev/shm, plain, .gz (one member) and bgzip'd, C2 parameters,
    wall clock from the first byte read to hulk_finish, on a context whose tables exist (the fastest of four runs, each on a
    fresh context; all four are in the line).  Host-bound (parse / inflate), reported beside the kernel-path figure, never as it."""
    import gzip
    import shutil
    import tempfile
    import hulk_amd
    from hulk_amd import synth
    d = tempfile.mkdtemp(prefix="hulk_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"reads": n_reads, "note": "hulk_sketch_files on a page-cached file, wall clock incl. parse, PCIe and hulk_finish"}
    try:
        plain = os.path.join(d, "reads.fq")
        qual = b"I" * READ_LEN
        with open(plain, "wb") as fh:
            for first in range(0, n_reads, 100_000):
                n = min(100_000, n_reads - first)
                bases, _ = synth.reads_numpy(first, n, READ_LEN)
                bb = bases[:n * READ_LEN].tobytes()
                fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (first + i, bb[i * READ_LEN:(i + 1) * READ_LEN], qual) for i in range(n)))
        gz = plain + ".gz"
        with open(plain, "rb") as fi, gzip.open(gz, "wb", compresslevel=1) as fo:
            shutil.copyfileobj(fi, fo, 1 << 24)
        import zlib
        bg = plain + ".bgzf.gz"                  # bgzip's container: 64 KiB members with their size in the header, inflated side by side
        with open(plain, "rb") as fi, open(bg, "wb") as fo:
            while True:
                piece = fi.read(65280)
                c = zlib.compressobj(1, zlib.DEFLATED, -15)
                body = c.compress(piece) + c.flush()
                fo.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\x00BC\x02\x00" + (18 + len(body) + 8 - 1).to_bytes(2, "little") + body +
                         (zlib.crc32(piece) & 0xffffffff).to_bytes(4, "little") + len(piece).to_bytes(4, "little"))
                if not piece:                    # (the empty member written last is the format's end-of-file marker)
                    break
        from hulk_amd import ingest
        for label, path in (("plain", plain), ("gz", gz), ("bgzf", bg)):
            runs = []
            for _ in range(4):                   # every run on a fresh context; all four listed, the fastest reported (the first of a
                g = hulk_amd.GpuSketcher(K, W, S, interval=INTERVAL)     # process also pays for its first threads and huge pages)
                t0 = time.perf_counter()
                st = g.sketch_files([path])
                g.finish()
                dt = time.perf_counter() - t0
                mins, _ = g.sketch()
                g.close()
                assert st["n_seqs"] == n_reads
                runs.append(dt)
            best = min(runs)
            _, _, pst = ingest.parse_files([path], collect=False)        # the host side alone: hulk_parse_files, no GPU sink
            out[label] = {"value": n_reads / best, "unit": "reads/s", "seconds": best, "seconds_all_runs": runs, "file_bytes": os.path.getsize(path),
                          "parse_only_reads_per_s": pst["n_seqs"] / pst["seconds"] if pst["seconds"] > 0 else None,
                          "sketch_md5": hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()}
        assert out["plain"]["sketch_md5"] == out["gz"]["sketch_md5"] == out["bgzf"]["sketch_md5"]
        # The plain file four times over (8 M reads, 2.5 GB of text): the line machine on the device (the default: the host
        # only read()s blocks into pinned memory and copies them over PCIe) beside the host's parser threads, same sketch.
        big = os.path.join(d, "reads_x4.fq")
        with open(big, "wb") as fo:
            for _ in range(4):
                with open(plain, "rb") as fi:
                    shutil.copyfileobj(fi, fo, 1 << 24)
        from hulk_amd import _lib as _l
        for label, flags in (("plain_8m", 0), ("plain_8m_host_parser", _l.HULK_INGEST_HOST_PARSER)):
            runs, md5 = [], None
            for _ in range(3):
                g = hulk_amd.GpuSketcher(K, W, S, interval=INTERVAL)
                t0 = time.perf_counter()
                st = g.sketch_files([big], opts={"flags": flags})
                g.finish()
                dt = time.perf_counter() - t0
                mins, _ = g.sketch()
                g.close()
                assert st["n_seqs"] == 4 * n_reads
                runs.append(dt); md5 = hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()
            out[label] = {"value": 4 * n_reads / min(runs), "unit": "reads/s", "reads": 4 * n_reads, "seconds": min(runs), "seconds_all_runs": runs,
                          "file_bytes": os.path.getsize(big), "file_GB_per_s": os.path.getsize(big) / min(runs) / 1e9, "sketch_md5": md5}
        assert out["plain_8m"]["sketch_md5"] == out["plain_8m_host_parser"]["sketch_md5"]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def c5_leg():
    """BASELINE configs[4] ("C5"): `hulk smash` over 1024 sketches of sketchSize 2048 (cmd/smash.go:183-226), both metrics:
    end to end through hulk_smash (host arrays in, PCIe both ways) and the distance kernel alone (hulk_smash_ex, HIP events).
    k_smash is VALU-bound: per slot a thread does 72 VALU instructions for its 4 x 4 pairs (16 v_cmp_eq_f64, 32 v_cndmask,
    20 v_add_f64; 48 for the plain Jaccard count) = 4.5 per (pair, slot) against 0.375 ds_read_b128 —
    valu_frac = N^2 * S * 4.5 / 64 wave-instructions * 4.4 cycles (profiles/r03_op_cost.txt) / 1024 SIMDs over the kernel time.
    `directory`: the reference's whole command (cmd/smash.go:160-226, sketchio.go:100-195): 1024 sketch JSON files are
    written to a scratch directory first (untimed), then hulk_amd.smash.smash() loads and MD5-verifies them, orders them,
    computes the matrix on the GPU and writes the CSV — timed as a whole and per stage."""
    from hulk_amd.smash import distance_matrix
    rng = np.random.default_rng(5)
    N, S_ = C5["n"], C5["S"]
    base = rng.integers(0, 194481, size=S_).astype(np.uint64)
    mins = np.where(rng.random((N, S_)) < 0.5, base, rng.integers(0, 194481, size=(N, S_)).astype(np.uint64))
    w = -rng.gamma(2.0, 1e-3, size=(N, S_))
    distance_matrix(mins[:8], w[:8], "weightedjaccard")           # module load
    out = {"workload": f"C5: pairwise matrix over {N} synthetic sketches, sketchSize {S_} (hulk smash, cmd/smash.go:183-226)",
           "pairs": N * N}
    for metric in ("weightedjaccard", "jaccard"):
        best = None
        per_pair_slot = 4.5 if metric == "weightedjaccard" else 3.0
        valu_cycles = N * N * S_ * per_pair_slot / 64.0 * 4.4 / 1024.0
        for _ in range(3):
            tm = {}
            t0 = time.perf_counter()
            d = distance_matrix(mins, w, metric, timing=tm)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, tm["kernel_ms"], hashlib.md5(np.ascontiguousarray(d).tobytes()).hexdigest())
        out[metric] = {"ms_end_to_end": best[0] * 1e3, "ms_kernel": best[1], "pairs_per_s": N * N / best[0],
                       "valu_frac": (valu_cycles / (CLOCK_GHZ * 1e9)) / (best[1] * 1e-3) if best[1] > 0 else None,
                       "matrix_md5": best[2]}
    # the command as the reference runs it: a directory of sketch files in, a CSV out
    import shutil
    import tempfile
    from hulk_amd import sketchio
    from hulk_amd import smash as smash_mod
    d_ = tempfile.mkdtemp(prefix="hulk_c5_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for i in range(N):
            hd = sketchio.HULKdata()
            hd.filename = f"sample_{i:04d}.fq"
            hd.add(sketchio.HistoSketch(21, mins[i], w[i], 194481, False))
            hd.write_json(os.path.join(d_, f"sample_{i:04d}.json"))
        file_bytes = sum(os.path.getsize(os.path.join(d_, f)) for f in os.listdir(d_))
        runs_ = []
        for _ in range(3):                 # the whole command three times, the fastest reported (all listed): a 40 ms region on a shared host
            st_ = {}
            t0 = time.perf_counter()
            order, dist = smash_mod.smash(d_, os.path.join(d_, "out"), ksize=21, algo="histosketch", metric="weightedjaccard", stages=st_)
            runs_.append((time.perf_counter() - t0, st_))
            assert len(order) == N
        total, stages = min(runs_, key=lambda r: r[0])
        t0 = time.perf_counter()                               # the same command with the loader and the CSV in Python (the round-5 form)
        smash_mod.smash_python(d_, os.path.join(d_, "out_py"), ksize=21, algo="histosketch", metric="weightedjaccard")
        total_py = time.perf_counter() - t0
        same_csv = open(os.path.join(d_, "out.hulk-matrix.csv"), "rb").read() == open(os.path.join(d_, "out_py.hulk-matrix.csv"), "rb").read()
        out["directory"] = {"files": N, "file_bytes": file_bytes, "seconds_total": total, "seconds_all_runs": [r[0] for r in runs_], "seconds_total_python_loader": total_py,
                            "csv_same_as_python_form": same_csv, "loader": "native (hulk_smash_files: JSON + MD5 on host threads, CSV by the library)",
                            "seconds_load_and_md5": stages.get("load"), "seconds_matrix": stages.get("matrix"),
                            "seconds_csv": stages.get("csv"),
                            "matrix_md5": hashlib.md5(np.ascontiguousarray(dist).tobytes()).hexdigest(),
                            "same_matrix_as_arrays": hashlib.md5(np.ascontiguousarray(dist).tobytes()).hexdigest() == out["weightedjaccard"]["matrix_md5"]}
    finally:
        shutil.rmtree(d_, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("sharded", "sliced-strong", "sliced-weak"), default="sharded",
                    help="N > 1: how the global stream is shared (see the module docstring); at N = 1 all three are the same "
                         "work and the plain single-GPU calls are used unless --force-collective")
    ap.add_argument("--no-prune", action="store_true",
                    help="the timed pass itself runs with the exact bounds of the CWS stage off (HULK_FLAG_NO_PRUNE): every "
                         "interval is evaluated against the whole table (profiling aid; implies --single-pass)")
    ap.add_argument("--n-frac", type=float, default=0.0,
                    help="variant workload: this fraction of the reads gets one 'N' at a pseudo-random position (real Illumina "
                         "data has such reads; they leave the table-free fast path of the minimizer kernel).  Not the headline.")
    ap.add_argument("--lanes", type=int, default=0, help="hulk_params.work_lanes of the timed contexts (0 = the library's default, 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the value_cold pass (C2 exactly: 10 M reads, no warm-up)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the FASTQ file -> sketch figures (N = 1 only)")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 leg (k=31, sketchSize=1024, decay; N = 1 only)")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 leg (hulk smash 1024 x 2048; N = 1 only)")
    ap.add_argument("--no-long-reads", action="store_true", help="skip the long-sequence leg (5 kb reads, 500 kb contigs; N = 1 only)")
    ap.add_argument("--no-c4", action="store_true", help="N > 1: skip value_c4 (50 M reads per rank on fresh contexts)")
    ap.add_argument("--no-long", action="store_true", help="skip the long pass (ms_per_step_long)")
    ap.add_argument("--single-pass", action="store_true",
                    help="headline and kernel durations only: no value_unpruned, no long pass, no other modes")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the sharded-step path (RCCL communicator, exchange inside the library) even at world size 1 (test aid)")
    ap.add_argument("--loopback", type=int, default=0, metavar="G",
                    help="projection aid at N = 1: time rank 0's share of a G-rank sharded step on this GPU, the other ranks' "
                         "contributions stood in for by copies of its own (hulk_comm_init_loopback); not a headline")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        self_spawn(args)

    # RCCL / the HIP runtime print banners on stdout; the contract is ONE JSON line there.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import hulk_amd
    from hulk_amd import _lib, synth
    from hulk_amd.distributed import gloo_exchange, interval_slice, num_steps, slot_shard, step_share

    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    transport = os.environ.get("HULK_BENCH_TRANSPORT", "rccl")         # "gloo": test aid, all ranks on GPU 0 (module docstring)
    if transport not in ("rccl", "gloo", "fakerccl"):
        raise SystemExit("HULK_BENCH_TRANSPORT must be rccl, gloo or fakerccl")
    shared_gpu = transport in ("gloo", "fakerccl")                     # test aids: every rank on GPU 0, torch.distributed over gloo
    if transport == "fakerccl":
        os.environ.setdefault("HULK_RCCL_LIB", os.path.join(ROOT, "tests", "cpp", "libfakerccl.so"))
        if not os.path.exists(os.environ["HULK_RCCL_LIB"]):
            raise SystemExit(f"HULK_BENCH_TRANSPORT=fakerccl: {os.environ['HULK_RCCL_LIB']} is missing (python -c 'import __graft_entry__ as g; g.build()')")
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    use_dist = world > 1 or args.force_collective
    loop_world = args.loopback if (args.loopback > 1 and world == 1) else 0
    rccl_ranks = 0
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
            assert dist.get_backend() == "nccl"
            rccl_ranks = dist.get_world_size()
        assert dist.get_world_size() == world

    def host_tensor(vals, dtype):
        return torch.tensor(vals, dtype=dtype, device="cpu" if shared_gpu else device)

    # ---- the line so far, and the watchdog that prints it if a leg never returns -------------------------------------
    out = {}                       # rank 0: the JSON line; filled as the legs complete
    state = {"leg": "headline", "kick": time.monotonic(), "printed": False, "headline_done": False}
    lock = threading.Lock()

    def print_line():
        with lock:
            if state["printed"]:
                return
            state["printed"] = True
            if rank == 0:
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                print(json.dumps(out), flush=True)
                os.dup2(2, 1)

    def watchdog():
        while not state["printed"]:
            time.sleep(1.0)
            if time.monotonic() - state["kick"] > LEG_TIMEOUT_S:
                leg_ = state["leg"]
                sys.stderr.write(f"bench.py rank {rank}: leg '{leg_}' has not returned for {LEG_TIMEOUT_S:.0f} s; giving up on it\n")
                if not state["headline_done"]:
                    os._exit(3)                              # nothing to print: the headline itself hangs
                out[f"{leg_}_error"] = f"timeout: no return within {LEG_TIMEOUT_S:.0f} s (watchdog)"
                print_line()
                os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    leg_errors = {}
    dist_ok = [True]               # collective legs go on only while every rank finished every collective leg so far

    def run_leg(name, fn, collective=False):
        """fn() fault-isolated: an exception becomes `<name>_error` in the line (and, for a leg that holds collectives, every
        later collective leg is skipped on every rank: the ranks agree on the outcome after each such leg)."""
        state["leg"], state["kick"] = name, time.monotonic()
        if collective and not dist_ok[0]:
            leg_errors[name] = "skipped: an earlier collective leg failed on some rank"
            return None
        res, err = None, None
        try:
            if name in FAIL_LEGS:
                raise RuntimeError(f"HULK_BENCH_FAIL={name} (test aid)")
            if name in HANG_LEGS:
                time.sleep(1e9)
            res = fn()
        except BaseException as e:                           # noqa: BLE001 — the line must survive anything a leg does
            if isinstance(e, KeyboardInterrupt):
                raise
            err = f"{type(e).__name__}: {e}"[:400]
            import traceback
            traceback.print_exc(file=sys.stderr)
        if collective and use_dist:
            state["kick"] = time.monotonic()
            f = host_tensor([0 if err else 1], torch.int32)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            if not bool(f.item()):
                dist_ok[0] = False
                err = err or "failed on another rank"
                res = None
        if err:
            leg_errors[name] = err
        state["kick"] = time.monotonic()
        return res

    steps, warmup = args.steps, args.warmup
    mode = args.mode
    shard_world = loop_world or world                  # ranks a sharded step is laid out for

    def share(m):
        """(reads per spectrum, reads of this rank per step, reads of the global stream per step) under mode m"""
        if m == "sharded":
            return INTERVAL, BATCH * INTERVAL, shard_world * BATCH * INTERVAL
        sc_ = "strong" if m == "sliced-strong" else "weak"
        per = interval_slice(sc_, 0, INTERVAL, rank, world)[1]
        return per, per * BATCH, (INTERVAL if sc_ == "strong" else INTERVAL * world) * BATCH

    per_interval, reads_per_rank_step, reads_per_step = share(mode)
    global_interval = INTERVAL * world if mode == "sliced-weak" else INTERVAL
    sb, sc = slot_shard(S, rank if not loop_world else 0, shard_world)

    stream = torch.cuda.Stream(device=device)          # torch's stream: only the synthetic input is generated on it (and synchronised
    torch.cuda.set_stream(stream)                      # before use); the contexts run on their own private streams

    def make_input(m, max_buf, n_total_steps=None):
        """this rank's share of every step under mode `m`, resident in HBM: (buffers, offsets, reads per spectrum, reads per
        step of this rank, mode) — `sharded`: its whole intervals of the step (distributed.step_share), `sliced-*`: its slice
        of every interval (distributed.interval_slice) — so an N-rank run sketches the same global stream as ONE rank"""
        nb = min(n_total_steps or (steps + warmup), max_buf)        # distinct steps kept in HBM (reused cyclically beyond that)
        per, n_step, _ = share(m)
        bufs = []
        for s_ in range(nb):
            parts = []
            if m == "sharded":
                first, cnt, _ = step_share(s_, BATCH, INTERVAL, rank, shard_world)
                b, _ = synth.reads_torch(first, cnt, READ_LEN, device=device)
                parts.append(b[:cnt * READ_LEN])
            else:
                for t in range(BATCH):
                    first, cnt = interval_slice("strong" if m == "sliced-strong" else "weak", s_ * BATCH + t, INTERVAL, rank, world)
                    b, _ = synth.reads_torch(first, cnt, READ_LEN, device=device)
                    parts.append(b[:cnt * READ_LEN])
            pad = torch.zeros(16, dtype=torch.uint8, device=device)
            sbuf = torch.cat(parts + [pad])
            if args.n_frac > 0:                   # one 'N' in a deterministic pseudo-random subset of the reads
                idx = torch.arange(n_step, dtype=torch.int64, device=device)
                hsh = ((idx + s_ * 1_000_003) * 0x9E3779B1) & 0xFFFFFFFF
                sel = idx[(hsh.double() / 4294967296.0) < args.n_frac]
                sbuf[sel * READ_LEN + (hsh[sel] >> 8) % READ_LEN] = ord("N")
            bufs.append(sbuf)
        offs = torch.arange(n_step + 1, dtype=torch.int64, device=device) * READ_LEN
        return bufs, offs, per, n_step, m

    main_input = make_input(mode, 24)
    torch.cuda.synchronize()
    ramp_buf = torch.zeros(1 << 26, dtype=torch.float32, device=device)      # 256 MB of scratch for the clock ramp (run_pass)

    rccl_error = [None]            # set on every rank when RCCL could not be bound / initialised on ANY rank (reported in the line)
    host_group = [None]

    def connect(sk, remake=None):
        """the context's communicator: RCCL (rank 0's id travels over torch.distributed), the host transport over gloo
        (test aid), or the loopback stand-in (--loopback).  Returns the context to use: when hulk_comm_init fails on any
        rank (librccl.so.1 missing, ncclCommInitRank refusing) all ranks agree on it, say so on stderr and in the JSON line
        (comm.transport), and the run goes on over the library's HOST transport on a gloo group — same protocol, same
        kernels, slower exchange — rather than leaving the scaling run without a number."""
        if loop_world:
            sk.comm_init_loopback(0, loop_world)
            return sk
        if transport == "gloo":
            sk.comm_init_host(rank, world, gloo_exchange(dist))
            return sk
        if rccl_error[0] is None:
            uid = err = None
            if rank == 0:
                try:
                    uid = hulk_amd.GpuSketcher.comm_unique_id()
                except _lib.HulkError as e:
                    err = str(e)
            ids = [uid]
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is None:
                err = err or "rank 0 could not create an RCCL unique id"
            else:
                try:
                    sk.comm_init(ids[0], rank, world)
                except _lib.HulkError as e:
                    err = str(e)
            errs = [None] * world
            dist.all_gather_object(errs, err)
            bad = [e for e in errs if e]
            if not bad:
                return sk
            rccl_error[0] = bad[0]
            host_group[0] = dist.new_group(backend="gloo")
            if rank == 0:
                print(f"bench.py: RCCL transport unavailable ({bad[0]}); using the library's host transport over gloo", file=sys.stderr)
            if err is None and remake is not None:             # this rank's context already holds a communicator
                sk.close()
                sk = remake()
        sk.comm_init_host(rank, world, gloo_exchange(dist, host_group[0]))
        return sk

    def run_pass(prune, inp=None, brackets=0, serial=False, n_steps=None):
        """warm-up + the timed K steps on a fresh context; prune=False disables the exact bounds of the CWS stage
        (HULK_FLAG_NO_PRUNE), so that every interval streams the whole table like the reference does; serial=True puts the
        flush on the work stream and bins a batch in one piece (HULK_FLAG_NO_OVERLAP: every kernel alone); brackets: the
        hulk_set_profiling mask of the timed steps (0 = none)."""
        bufs, offs, per, n_step, in_mode = inp if inp is not None else main_input
        k_steps = n_steps or steps
        comm = use_dist or loop_world
        sharded = comm and in_mode == "sharded"
        flags = (0 if prune else _lib.HULK_FLAG_NO_PRUNE) | (_lib.HULK_FLAG_NO_OVERLAP if serial else 0)
        def make():
            return hulk_amd.GpuSketcher(K, W, S, interval=INTERVAL if (sharded or not comm) else 0, decay_ratio=1.0, device=dev_index,
                                        slot_begin=sb, slot_count=sc, flags=flags, batch=BATCH, work_lanes=args.lanes)
        sk = make()
        assert sk.batch_size == BATCH
        if comm:
            sk = connect(sk, make)

        def one_step(t):
            b = bufs[t % len(bufs)]
            if sharded:
                sk.step_sharded(b.data_ptr(), offs.data_ptr(), n_step, READ_LEN, b.numel(), shard_world * BATCH)
            elif comm:
                sk.step_sliced(b.data_ptr(), offs.data_ptr(), n_step, READ_LEN, b.numel(), per, BATCH)
            else:                                  # one GPU: the product's entry point, the interval rule inside (16 flushes per call)
                sk.add_reads_device(b.data_ptr(), offs.data_ptr(), n_step, READ_LEN, b.numel())

        # Creating a context leaves the GPU idle for 10-90 ms (allocations, the CWS tables from their cache) and its clocks
        # drop: the W warm-up steps that follow (5 ms at the driver's W = 5) are over before they are back, and the first ~15
        # timed steps run 2-15 % slow (profiles/r03_firstpass_steps.txt; 0.44-1.5 ms of a 20-step pass, by the box).  RAMP_MS
        # of plain elementwise kernels on a scratch tensor put the chip back under load first; the W warm-up steps and the K
        # timed steps follow without a gap.  (Nothing is taken out of the timed region: `ms_per_step_long` is the same
        # measurement over 200 steps.  A second hulk context as the load was tried first: its allocations and frees made
        # later contexts of the process 20 % slower — profiles/r04_bench_ramp.txt.)
        if RAMP_MS > 0 and not serial:
            t_end = time.perf_counter() + RAMP_MS * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(8):
                    ramp_buf.sin_()
                torch.cuda.synchronize()
        for t in range(warmup):
            one_step(t)
        sk.synchronize()
        torch.cuda.synchronize()
        sk.set_profiling(brackets)             # hulk_set_profiling: 0 none, 1 = all instrumented kernels, else a mask
        tiles0 = sk.scan_stats()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stamps = []                            # HULK_BENCH_STEPTIMES (diagnosis): 1 = when each call returned, 2 = with a sync per step
        for t in range(warmup, warmup + k_steps):
            one_step(t)
            state["kick"] = time.monotonic()
            if STEPTIMES:
                if STEPTIMES == 2:
                    sk.synchronize()
                stamps.append(time.perf_counter())
        if STEPTIMES and rank == 0:
            sys.stderr.write("step times (ms, %s): " % ("synchronised" if STEPTIMES == 2 else "host call returns") +
                             " ".join("%.3f" % ((b - a) * 1e3) for a, b in zip([t0] + stamps[:-1], stamps)) + "\n")
        sk.synchronize()                       # (queues the last step's flush, which otherwise waits for a next batch)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        tiles1 = sk.scan_stats()
        prof = {k: sk.get_profile(k) for k in ("k_cws_scan", "k_minimizer_fast", "k_jump_bin", "k_jump_left")} if brackets else {}
        table = sk.profile_table() if (brackets and brackets != 1 and (brackets & 32)) else None     # every launch of the chain (hulk_get_profile_table)
        sk.set_profiling(False)
        if use_dist:
            tt = host_tensor([elapsed], torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())

        sk.finish()
        counters = sk.counters()
        mins, weights = sk.gather_sketch() if (comm and not loop_world) else sk.sketch()
        cstats = sk.comm_stats() if comm else None
        sk.close()
        return dict(elapsed=elapsed, steps=k_steps, prof=prof, counters=counters, mins=mins, weights=weights, tiles0=tiles0,
                    tiles1=tiles1, comm=cstats, table=table)

    def run_cold():
        """C2 exactly as BASELINE.json states it: 10 M reads, interval 100k, through the interval rule of
        hulk_add_reads_device on a FRESH context (the first batch evaluates the whole CWS table), no warm-up; the clock
        stops after hulk_finish (final flush + device error check).  Context creation (CWS table generation, the
        reference pays it once at pipeline/sketch.go:277) is timed separately."""
        chunk = INTERVAL * BATCH
        chunks = []
        for first in range(0, C2_READS, chunk):
            n = min(chunk, C2_READS - first)
            b, off = synth.reads_torch(first, n, READ_LEN, device=device)
            chunks.append((b, off, n))
        torch.cuda.synchronize()
        def ramp():
            t_end = time.perf_counter() + RAMP_MS * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(8):
                    ramp_buf.sin_()
                torch.cuda.synchronize()
        runs, ramped = [], []
        for rep in range(4):          # two complete cold runs as BASELINE states C2 — each on its own fresh context; the faster one is
            t0 = time.perf_counter()  # reported (both listed): the timed region is 7 ms, and one host hiccup on a shared box is 100x that —
            sk = hulk_amd.GpuSketcher(K, W, S, interval=INTERVAL, decay_ratio=1.0, device=dev_index, batch=BATCH, work_lanes=args.lanes)
            torch.cuda.synchronize()  # and two more with the headline's RAMP_MS of elementwise load between context creation and the first
            if rep >= 2:              # read (`value_cold_ramped`): creating a context idles the GPU for 10-90 ms and its clocks drop, so the
                ramp()                # plain figure measures the box's idle clocks as much as the code; the ramped one separates the two
            t1 = time.perf_counter()
            for b, off, n in chunks:
                sk.add_reads_device(b.data_ptr(), off.data_ptr(), n, READ_LEN, b.numel())
            sk.finish()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            mins, _ = sk.sketch()
            sk.close()
            (runs if rep < 2 else ramped).append((t2 - t1, t1 - t0, hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()))
        assert len({r[2] for r in runs + ramped}) == 1
        best, best_r = min(runs), min(ramped)
        return {"value_cold": C2_READS / best[0], "cold_seconds": best[0], "cold_create_seconds": best[1],
                "cold_seconds_all_runs": [r[0] for r in runs], "cold_reads": C2_READS, "cold_sketch_md5": best[2],
                "value_cold_ramped": C2_READS / best_r[0], "cold_ramped_seconds_all_runs": [r[0] for r in ramped],
                "cold_ramped_note": f"the same fresh-context 10 M-read run with {RAMP_MS:g} ms of elementwise torch kernels between hulk_create and the first read "
                                    "(the clock starts after the ramp): clocks up, everything else cold — the first batch still evaluates the whole CWS table"}

    def run_c3():
        """BASELINE configs[2] ("C3") on an HBM-resident sample: k = 31, sketchSize = 1024, concept drift on (decay 0.02),
        interval 100k — 22.7 GB of fp64 CWS tables + 3.8 GB of K; one warm-up batch, then C3["reads"] reads through the
        interval rule of hulk_add_reads_device, wall clock between two hulk_synchronize; then the same again with every kernel alone
        and k_minimizer_fast / k_cmsd_freq (the count-min replay with decay, countmin.go:141-147) bracketed."""
        k3, w3, S3, I3 = C3["k"], C3["w"], C3["S"], C3["interval"]
        stp = I3 * BATCH
        nbuf = min(4, (C3["reads"] + stp - 1) // stp)
        bufs = [synth.reads_torch(s_ * stp, stp, READ_LEN, device=device) for s_ in range(nbuf)]
        torch.cuda.synchronize()
        res = {"workload": f"C3: synthetic 150bp reads, k={k3}, w={w3}, sketchSize={S3}, decay {C3['decay']} (concept drift), "
                           f"interval={I3}, {BATCH} intervals per batch, HBM-resident input (4 buffers of {stp} reads in turn), {C3['reads']} reads "
                           "timed after one warm-up batch, the clock stopped after the last batch's flush"}
        for label, serial in (("overlapped", False), ("kernels_alone", True)):
            t0 = time.perf_counter()
            sk = hulk_amd.GpuSketcher(k3, w3, S3, interval=I3, decay_ratio=C3["decay"], device=dev_index,
                                      batch=BATCH, work_lanes=args.lanes, flags=_lib.HULK_FLAG_NO_OVERLAP if serial else 0)
            torch.cuda.synchronize()
            create_s = time.perf_counter() - t0
            b, o = bufs[0]
            sk.add_reads_device(b.data_ptr(), o.data_ptr(), stp, READ_LEN, b.numel())
            sk.synchronize(); torch.cuda.synchronize()
            if serial:
                sk.set_profiling(32)                       # every launch of the chain timed (hulk_get_profile_table): steady state, the warm-up batch is out
            state["kick"] = time.monotonic()
            t1 = time.perf_counter()
            done, i = 0, 1
            while done < C3["reads"]:
                b, o = bufs[i % nbuf]
                n = min(stp, C3["reads"] - done)           # (the last call: what is left of the 50 M)
                sk.add_reads_device(b.data_ptr(), o.data_ptr(), n, READ_LEN, b.numel())
                done += n; i += 1
            sk.synchronize()                       # (private streams: the context's own synchronisation point stops the clock)
            ms = (time.perf_counter() - t1) * 1e3
            if serial:
                tbl = sk.profile_table()
                nb = done / stp
                res["kernels_alone"] = {"ms_per_batch": ms / nb, "reads_per_s": done / ms * 1e3, "reads": done, "batches": nb,
                                        **{kk + "_us": (tbl[kk][1] / max(tbl[kk][0], 1)) * 1e3 for kk in ("k_minimizer_fast", "k_jump_bin", "k_jump_left", "k_cmsd_freq") if kk in tbl},
                                        # every kernel of the binning chain and of the decay flush: microseconds per batch (all its launches), steady state
                                        "us_per_batch": {kk: round(v[1] / nb * 1e3, 2) for kk, v in sorted(tbl.items(), key=lambda kv: -kv[1][1])},
                                        "launches_per_batch": {kk: round(v[0] / nb, 2) for kk, v in tbl.items()},
                                        "us_per_batch_sum": round(sum(v[1] for v in tbl.values()) / nb * 1e3, 1),
                                        "note": "HULK_FLAG_NO_OVERLAP: one stream, one piece per batch; hulk_set_profiling(32): an event in front of every launch, "
                                                "a kernel = the time to the next event (includes ~2-4 us of event handling per launch); the first (cold) batch excluded"}
                sk.set_profiling(0)
            else:
                res.update({"value": done / ms * 1e3, "unit": "reads/s", "reads": done, "ms_per_batch": ms / (done / stp),
                            "create_seconds": create_s})
            tiles = sk.scan_stats()
            sk.finish()
            mins, wts = sk.sketch()
            if not serial:
                res.update({"scan_tiles_read": tiles[0], "scan_tiles_covered": tiles[1], "negative_weights": int((wts < 0).sum()),
                            "sketch_md5": hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()})
            else:
                res["kernels_alone"]["sketch_md5"] = hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()
            sk.close()
            torch.cuda.empty_cache()
        assert res["sketch_md5"] == res["kernels_alone"]["sketch_md5"], "C3: the one-stream run disagrees"
        return res

    def run_long_reads():
        """Reads the short-read kernels do not take (> 512 bases) and FASTA contigs (src/pipeline/sketch.go:102-135: a '>' record's
        lines concatenated into ONE sequence) go to k_minimizer_bin (<= 1024 k-mer positions) and the grouped long-sequence
        kernel k_long_tile (hulk_minimizer.hip; until round 6 k_long_hash + k_long_emit).  Two shapes, k = 21, w = 9, sketchSize = 512, HBM-resident, ONE
        spectrum (interval 0): 200 k reads x 5 kb and 2 k contigs x 500 kb = 1 Gbase each.  Per shape: the second of two identical
        calls is timed (the first sizes the grow-only scratch), binning only (hulk_synchronize stops the clock) and with the one
        flush of hulk_finish; then the same call on a one-stream context with every launch timed (hulk_set_profiling(32))."""
        res = {"workload": "long sequences: synthetic ACGT, k=21, w=9, sketchSize=512, interval 0 (one spectrum), HBM-resident; "
                           "reference path: pipeline/sketch.go:102-135 (FASTA) / any read beyond the fast kernel's 512 bases"}
        for label, n, L in (("reads_5kb", 200_000, 5_000), ("contigs_500kb", 2_000, 500_000)):
            b, off = synth.reads_torch(0, n, L, device=device)
            torch.cuda.synchronize()
            shape = {"sequences": n, "length": L, "bases": n * L}
            for serial in (False, True):
                sk = hulk_amd.GpuSketcher(K, W, S, interval=0, device=dev_index, flags=_lib.HULK_FLAG_NO_OVERLAP if serial else 0)
                sk.add_reads_device(b.data_ptr(), off.data_ptr(), n, L, b.numel())
                sk.synchronize(); torch.cuda.synchronize()
                if serial:
                    sk.set_profiling(32)
                state["kick"] = time.monotonic()
                t0 = time.perf_counter()
                sk.add_reads_device(b.data_ptr(), off.data_ptr(), n, L, b.numel())
                sk.synchronize()
                dt = time.perf_counter() - t0
                if serial:
                    tbl = sk.profile_table()
                    sk.set_profiling(0)
                    shape["kernels_alone"] = {"seconds": dt, "us": {kk: round(v[1] * 1e3, 1) for kk, v in sorted(tbl.items(), key=lambda kv: -kv[1][1])},
                                              "launches": {kk: v[0] for kk, v in tbl.items()}}
                    hk = tbl.get("k_long_tile")
                    if hk and hk[1] > 0:
                        # algorithmic bytes: the bases, once (SURVEY 8d's bin-side figure: L per read) — which is also all the kernel streams: a
                        # workgroup stages its tile's bases into LDS and hashes, window minima and run starts never leave it; what it adds are
                        # the atomics of the per-sequence set (one 64-bit CAS per run start, ~0.2 per position) and of the spectrum
                        shape["roofline_k_long_tile"] = {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achieved": n * L / (hk[1] * 1e-3) / 1e9,
                                                         "frac": n * L / (hk[1] * 1e-3) / 1e9 / 8000.0, "alg_bytes": n * L,
                                                         "total_us": hk[1] * 1e3, "launches": hk[0],
                                                         "note": "atomics-bound, not HBM-bound: ~0.4 device-scope atomics per position (set CAS + spectrum add) at "
                                                                 "the ~27 G/s this chip sustains are 15 of the kernel's 20 ms per Gbase"}
                else:
                    t1 = time.perf_counter()
                    sk.finish()
                    shape.update({"seconds_binning": dt, "reads_per_s": n / dt, "bases_per_s": n * L / dt,
                                  "seconds_with_final_flush": dt + (time.perf_counter() - t1)})
                    cnt = sk.counters()
                    assert cnt["n_reads"] == 2 * n and cnt["total_len"] == 2 * n * L, cnt
                    shape["minimizers_per_kb"] = cnt["n_minimizers"] / (2 * n * L / 1e3)
                    mins, _ = sk.sketch()
                    shape["sketch_md5"] = hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()
                sk.close()
            res[label] = shape
            del b, off
            torch.cuda.empty_cache()
        # ... and the reference's --fasta mode end to end (sketch.go:102-135: the lines of a '>' record concatenated, parsing stops at the
        # first empty line): a file of 200 contigs x 500 kb in 60-column lines on /dev/shm -> hulk_sketch_files(fasta) -> hulk_finish; the
        # fastest of three runs, each on a fresh context.  Host-bound (the FASTA line pump runs on the host's parser threads).
        import shutil
        import tempfile
        d_ = tempfile.mkdtemp(prefix="hulk_fa_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            nfa, Lfa = 200, 500_000
            path = os.path.join(d_, "contigs.fa")
            with open(path, "wb") as fh:
                for i in range(nfa):
                    seq = synth.reads_numpy(i, 1, Lfa)[0].tobytes()
                    fh.write(b">contig_%d\n" % i + b"\n".join(seq[j:j + 60] for j in range(0, Lfa, 60)) + b"\n")
            # the line pump on the device (hulk_fastq.hip k_fa_*: the default) and on the host's parser threads (HULK_INGEST_HOST_PARSER)
            runs, runs_host, md5s = [], [], set()
            for flags, into in ((0, runs), (_lib.HULK_INGEST_HOST_PARSER, runs_host)):
                for _ in range(3):
                    sk = hulk_amd.GpuSketcher(K, W, S, interval=0, device=dev_index)
                    state["kick"] = time.monotonic()
                    t0 = time.perf_counter()
                    st = sk.sketch_files([path], fasta=True, opts={"flags": flags})
                    sk.finish()
                    into.append(time.perf_counter() - t0)
                    assert st["n_seqs"] == nfa and st["total_len"] == nfa * Lfa, st
                    mins, _ = sk.sketch()
                    md5s.add(hashlib.md5(mins.astype("<u8").tobytes()).hexdigest())
                    sk.close()
            assert len(md5s) == 1
            # the same contigs through the device-pointer call: the file path changes nothing but where the bytes come from
            b, off = synth.reads_torch(0, nfa, Lfa, device=device)
            torch.cuda.synchronize()
            sk = hulk_amd.GpuSketcher(K, W, S, interval=0, device=dev_index)
            sk.add_reads_device(b.data_ptr(), off.data_ptr(), nfa, Lfa, b.numel())
            sk.finish()
            mins, _ = sk.sketch()
            sk.close()
            md5_file = md5s.pop()
            res["fasta_file"] = {"contigs": nfa, "length": Lfa, "file_bytes": os.path.getsize(path), "seconds": min(runs), "seconds_all_runs": runs,
                                 "bases_per_s": nfa * Lfa / min(runs), "file_GB_per_s": os.path.getsize(path) / min(runs) / 1e9, "sketch_md5": md5_file,
                                 "parser": "device (k_fa_*: the host read()s blocks into pinned memory)",
                                 "host_parser": {"seconds": min(runs_host), "seconds_all_runs": runs_host, "bases_per_s": nfa * Lfa / min(runs_host)},
                                 "same_sketch_as_device_buffers": hashlib.md5(mins.astype("<u8").tobytes()).hexdigest() == md5_file}
        finally:
            shutil.rmtree(d_, ignore_errors=True)
        return res

    def run_c4():
        """BASELINE configs[3] ("C4"), scaled to the ranks present: C4_READS_PER_RANK (50 M) x N reads of the global stream
        — 400 M at N = 8 — through hulk_step_sharded on FRESH contexts: no warm-up, the first step exchanges and evaluates
        the spectra against the whole CWS table, the last step is ragged, and the clock stops after hulk_finish and the
        EOF all-gather of the sketch (hulk_gather_sketch).  Context + communicator creation are timed separately."""
        total = C4_READS_PER_RANK * world
        ns = num_steps(total, BATCH, INTERVAL, world)
        chunks = []
        for s_ in range(ns):
            first, n, si = step_share(s_, BATCH, INTERVAL, rank, world, total)
            b, off = synth.reads_torch(first, max(n, 1), READ_LEN, device=device)
            chunks.append((b, off, n, si))
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        def make():
            return hulk_amd.GpuSketcher(K, W, S, interval=INTERVAL, decay_ratio=1.0, device=dev_index, slot_begin=sb,
                                        slot_count=sc, batch=BATCH, work_lanes=args.lanes)
        sk = connect(make(), make)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for b, off, n, si in chunks:
            sk.step_sharded(b.data_ptr(), off.data_ptr(), n, READ_LEN, b.numel(), si)
            state["kick"] = time.monotonic()
        sk.finish()
        mins, _ = sk.gather_sketch()
        torch.cuda.synchronize()
        dist.barrier()
        t2 = time.perf_counter()
        tt = host_tensor([t2 - t1], torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cs = sk.comm_stats()
        sk.close()
        del chunks
        torch.cuda.empty_cache()
        return {"value_c4": total / dt, "c4_seconds": dt, "c4_create_seconds": t1 - t0, "c4_reads": total, "c4_steps": ns,
                "c4_workload": f"C4: {total} synthetic 150bp reads sharded by whole intervals over {world} rank(s), k=21, w=9, "
                               f"sketchSize=512, interval={INTERVAL} of the global stream, fresh contexts, ragged last step, "
                               "clock stopped after the EOF gather",
                "c4_exchange": cs, "c4_sketch_md5": hashlib.md5(mins.astype("<u8").tobytes()).hexdigest()}

    # ------------------------------------------------------------------------------------------------------------------
    # The first pass of a process measures ~7 % low whatever runs before it short of a pass itself: it is the clock ramp of a
    # GPU leaving idle (HULK_BENCH_STEPTIMES=2 shows the first pass descending from 1.25 to 1.11 ms per synchronised step over
    # its 20 steps, later passes from 1.17 over their first 15; profiles/r03_firstpass_steps.txt).  So one pass is run and
    # discarded before the timed one ...
    run_pass(not args.no_prune)
    state["kick"] = time.monotonic()
    # ... and PREWARM_S seconds of discarded passes on top: on a box whose GPU has been idle (a fresh lease) the first process
    # otherwise measures 2-4 % below the ones after it (1.025 vs 0.988 ms per step; with 3 s of load first: 1.000 vs 0.990)
    t_pw = time.perf_counter() + PREWARM_S

    def more_prewarm():
        go = time.perf_counter() < t_pw
        if use_dist:                                   # every rank must run the same number of passes (they hold collectives)
            f = host_tensor([1 if go else 0], torch.int32)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            go = bool(f.item())
        return go

    while more_prewarm():
        run_pass(not args.no_prune)
        state["kick"] = time.monotonic()
    if os.environ.get("HULK_BENCH_REPEAT"):           # diagnosis: the same pass several times, ms per step of each on stderr
        for i in range(int(os.environ["HULK_BENCH_REPEAT"])):
            r = run_pass(not args.no_prune)
            sys.stderr.write(f"repeat {i}: {r['elapsed'] / steps * 1e3:.4f} ms/step\n")
    # ---- the HEADLINE: W warm-up + K timed steps, no event brackets (every bracket costs the stream two event records:
    # k_minimizer_fast's alone was 2 % of a step)
    main_pass = run_pass(not args.no_prune, brackets=0)
    elapsed, counters = main_pass["elapsed"], main_pass["counters"]
    mins, weights = main_pass["mins"], main_pass["weights"]
    total_reads = steps * reads_per_step
    value = total_reads / elapsed
    scaling = "strong" if mode == "sliced-strong" else "weak"
    comm_desc = None
    if use_dist or loop_world:
        how = ("loopback stand-in (no peers)" if loop_world else
               f"host transport over gloo (RCCL unavailable: {rccl_error[0]})" if rccl_error[0] else
               "RCCL (ncclAllGather / ncclAllReduce, bound by hulk_comm_init)" if transport == "rccl" else
               "the library's RCCL branch over the test double tests/cpp/libfakerccl.so (test aid: all ranks on GPU 0)" if transport == "fakerccl" else
               "host transport over gloo (test aid: all ranks on GPU 0)")
        what = ("one all-gather per step: k-mer spectra while an element can still lower a weight, count-min increments after"
                if mode == "sharded" else "one all-reduce (uint32 sum) of the step's spectra")
        comm_desc = {"inside": "libhulkhip.so (hulk_step_sharded / hulk_step_sliced)", "transport": how, "per_step": what,
                     "timed_pass": main_pass["comm"]}
    out.update({
        "metric": "reads/sec (150bp, k=21, sketch=512)", "value": value, "unit": "reads/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "C2: synthetic 150bp reads, k=21, w=9, sketchSize=512, "
                               f"interval={global_interval} reads of the global stream, "
                               f"{BATCH} intervals per rank and step ({reads_per_rank_step} reads; {reads_per_step} of the "
                               "global stream per step), HBM-resident input"
                               + (f", LOOPBACK: rank 0's share of a {loop_world}-rank sharded step, no peers" if loop_world else "")
                               + (", CWS-scan bounds OFF (--no-prune)" if args.no_prune else "")
                               + (f", VARIANT: {args.n_frac:g} of the reads carry one N" if args.n_frac > 0 else ""),
                   "reads_per_step": reads_per_step, "reads_per_rank_step": reads_per_rank_step,
                   "total_reads": total_reads, "intervals_per_step": BATCH, "global_interval": global_interval,
                   "mode": mode, "work_lanes": args.lanes or 2,
                   "split": ("whole intervals per rank" if mode == "sharded" else "a slice of every interval per rank"),
                   "parallelism": f"read-shard x{world}, replicated count-min, slot-sharded CWS"},
        "scaling_note": ("per-rank work per step is fixed (16 whole intervals = 1.6 M reads), the global interval stays the "
                         "reference's 100k reads: the sketch is the single-GPU sketch of the same (N times longer) stream"
                         if mode == "sharded" else "total work per step fixed" if mode == "sliced-strong" else
                         "per-rank work fixed, global interval N x 100k: another sketch than C2's"),
        "rccl_ranks": rccl_ranks, "prewarm_seconds": PREWARM_S, "ramp_ms": RAMP_MS,
        "collective": comm_desc,
        "roofline": None,          # (filled by the `kernels` leg)
        "sketch_md5": hashlib.md5(mins.astype("<u8").tobytes()).hexdigest(),
        "n_minimizers_rank0": counters["n_minimizers"],
        "timed_pass_note": "no HIP-event brackets in the timed steps; consecutive batches are binned on two alternating work streams "
                           "and flushed on a third, all private to the context (the library's default); per-kernel durations: the `kernels` leg; "
                           f"{RAMP_MS:g} ms of elementwise torch kernels run right before the W warm-up steps (creating the timed context leaves "
                           "the GPU idle and its clocks drop), outside the timed region",
    })
    # SURVEY.md §8(d) prices the path at L + 4*S*k^4/I bytes per read (one K pass per interval): a MODEL of the
    # reference's data movement, not traffic this implementation generates (one K pass serves BATCH intervals and
    # the exact bounds skip most of it) — kept for comparison with the survey's 1.94e9 reads/s/GPU figure only.
    model_bytes = READ_LEN + 4.0 * S * (K ** 4) / global_interval
    out["survey_model"] = {"bytes_per_read": model_bytes, "reads_per_s_at_hbm_peak": HBM_PEAK_GBS * 1e9 * world / model_bytes,
                           "value_over_model": value * model_bytes / 1e9 / (HBM_PEAK_GBS * world)}
    state["headline_done"] = True

    # ---- leg `kernels`: the same steps, every kernel alone (one stream, one piece per batch) and bracketed by HIP events on
    # the stream it runs on: the launch durations the roofline objects are computed from
    instr_pass = run_leg("kernels", lambda: run_pass(not args.no_prune, brackets=1, serial=True), collective=True)

    def add_rooflines():
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
        except Exception:
            pass
        isa_mix = {}
        try:
            isa_mix = json.load(open(os.path.join(ROOT, ISA_MIX)))
        except Exception:
            pass

        def from_profile(kernel, key):
            return pmc.get(kernel, {}).get(key) if world == 1 else None

        prof = instr_pass["prof"]
        tiles0, tiles1 = instr_pass["tiles0"], instr_pass["tiles1"]
        # Dominant kernel by time = k_minimizer_fast (K1a: bases -> distinct minimizers per read).  Its algorithmic
        # bytes are SURVEY.md §8(d)'s per-read figure for the bin side, L + 8 (one ASCII byte per base + the read's
        # offset), x the reads of one launch.  The minimizer list it hands to k_jump_bin (8 B value + 1 B spectrum slot
        # per distinct minimizer) is an artefact of this implementation: reported as intermediate_bytes, not priced.
        n_k1, k1_ms = prof["k_minimizer_fast"]
        k1_avg_s = (k1_ms / 1e3) / max(n_k1, 1)
        per_read_min = counters["n_minimizers"] / float(max(counters["n_reads"], 1))
        k1_bytes = float(reads_per_rank_step) * (READ_LEN + 8)
        k1_ach = k1_bytes / k1_avg_s / 1e9 if k1_avg_s > 0 else 0.0
        n_kj, kj_ms = prof["k_jump_bin"]
        kj_avg_s = (kj_ms / 1e3) / max(n_kj, 1)
        n_kl, kl_ms = prof["k_jump_left"]
        kl_avg_s = (kl_ms / 1e3) / max(n_kl, 1)
        longest = "k_minimizer_fast" if k1_avg_s >= kj_avg_s else "k_jump_bin"

        def valu_roofline(kernel, avg_s):
            """VALU-issue roofline of a launch.  SQ_INSTS_VALU (wave64 VALU instructions per launch) comes from the rocprofv3 PMC
            pass of this command (PMC_PROFILE); what an instruction costs was measured two independent ways
            (profiles/r03_op_cost.txt: whole launches by HIP events at the nominal clock, tools/ubench/op_cost.hip; every wave
            timing its own block with s_memtime — shader-clock ticks — grouped by the SIMD it ran on, op_cost2.hip):
            simple VOP2 ops (v_add/sub_u32, v_and/or/xor_b32, v_mov_b32, v_lshrrev_b32, v_mul/add_f32) issue in 2.35 cycles,
            everything else these kernels use (VOP3 integer ops, v_lshlrev_b32, min/max, DPP, cmp/cndmask, every 64-bit, fp64
            and packed op) in 4.4, v_rcp_f64 in 16.5.  MI355X_MICROARCH.md's 2 cycles is the first class only.
            floor_us      = SQ_INSTS_VALU / 1024 SIMDs x 2 cycles / 2.4 GHz   (every instruction priced as the fast class at the
                            nominal clock: an optimistic bound)
            floor_us_mix  = SQ_INSTS_VALU x (mean cycles of the kernel's own static instruction mix, tools/isa_mix.py ->
                            ISA_MIX) / 1024 / the measured shader clock under load (2.3 GHz)
            frac / frac_mix = floor / the launch duration measured live in this run."""
            insts = from_profile(kernel, "SQ_INSTS_VALU")
            if not insts or avg_s <= 0:
                return None
            per_launch = float(pmc.get("reads_per_launch", INTERVAL * BATCH))
            scale = reads_per_rank_step / per_launch
            floor_us = insts * scale / SIMDS * VALU_CYCLES / (CLOCK_GHZ * 1e3)
            out_ = {"kernel": kernel, "wave_instr_per_read": insts / per_launch,
                    "cycles_per_instr_assumed": VALU_CYCLES, "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
                    "floor_us": floor_us, "avg_launch_us": avg_s * 1e6, "frac": floor_us / (avg_s * 1e6),
                    "instr_from_profile": PMC_PROFILE,
                    "formula": "floor_us = SQ_INSTS_VALU / 1024 SIMDs * 2 cycles / 2.4 GHz; floor_us_mix = SQ_INSTS_VALU * "
                               "mean_cycles_of_the_static_mix / 1024 / 2.3 GHz (measured clock); frac = floor / avg_launch_us"}
            mixk = isa_mix.get(kernel)
            if mixk and mixk.get("mean_cycles_loop"):
                mc = float(mixk["mean_cycles_loop"])
                fm = insts * scale * mc / SIMDS / (CLOCK_MEASURED_GHZ * 1e3)
                out_.update({"cycles_per_instr_mix": mc, "mix": mixk.get("mix_loop"), "mix_from": ISA_MIX,
                             "op_costs_from": "profiles/r03_op_cost.txt", "clock_measured_ghz": CLOCK_MEASURED_GHZ,
                             "floor_us_mix": fm, "frac_mix": fm / (avg_s * 1e6)})
            return out_

        # The HBM-streaming kernel of the path = k_cws_scan.  Unpruned it makes ONE fp32 pass over this rank's
        # slice of K per launch (4*slots*k^4 bytes) + the BATCH reciprocal vectors; with the exact
        # bound test (no concept drift) it only reads the 8-slot x 256-bin tiles that can still lower a weight,
        # so the bytes it is priced on are the tiles it actually read (hulk_get_scan_stats) + the small tables.
        n_launch, scan_ms = prof["k_cws_scan"]
        wtiles = ((K ** 4 + 1023) // 1024) * 4
        full_bytes = 4.0 * sc * (K ** 4) + 4.0 * BATCH * (K ** 4)
        visited = (tiles1[0] - tiles0[0]) / max(n_launch, 1)
        covered = (tiles1[1] - tiles0[1]) / max(n_launch, 1)
        alg_bytes = visited * 8 * 256 * 4.0 + 4.0 * BATCH * (K ** 4) + 4.0 * sc * wtiles + 8.0 * BATCH * wtiles
        avg_s = (scan_ms / 1e3) / max(n_launch, 1)
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        out.update({
            "roofline": {"bound": "hbm", "kernel": "k_minimizer_fast", "achieved": k1_ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1_ach / HBM_PEAK_GBS,
                         "traffic": from_profile("k_minimizer_fast", "hbm_bytes_per_launch"),       # PMC passes of the same command, committed (traffic_profile)
                         "traffic_from_profile": from_profile("k_minimizer_fast", "hbm_bytes_per_launch"),
                         "traffic_profile": PMC_PROFILE if pmc else None,
                         "traffic_note": "HBM bytes per launch from the rocprofv3 PMC passes of this same command, committed as traffic_profile (collected "
                                         "separately per MI355X_MICROARCH.md: a --pmc run cannot be combined with the timed one); not a measurement of THIS run",
                         "launches": int(n_k1), "avg_launch_us": k1_avg_s * 1e6,
                         "alg_bytes_per_launch": k1_bytes, "alg_bytes_per_read": READ_LEN + 8,
                         "intermediate_bytes": float(reads_per_rank_step) * 9.0 * per_read_min,
                         "longest_kernel": longest,
                         "durations_from": "the `kernels` leg: the timed steps repeated on a fresh context with "
                                           "HULK_FLAG_NO_OVERLAP (one stream, a batch binned in ONE piece: every kernel runs "
                                           "alone) and every instrumented kernel bracketed by HIP events on the stream it is "
                                           "launched on — the headline pass itself carries no brackets and overlaps its kernels, "
                                           "which stretches each kernel's own duration; rocprofv3 of HULK_NO_OVERLAP=1 bench.py "
                                           "(profiles/r06_kernel_stats_serial.md) shows the same per-launch figures",
                         "note": f"single kernels by measured time: k_minimizer_fast {k1_avg_s * 1e6:.1f} us, k_jump_bin "
                                 f"{kj_avg_s * 1e6:.1f} us, k_jump_left {kl_avg_s * 1e6:.1f} us per launch of {reads_per_rank_step} reads "
                                 "(stage K1b = the last two; it has no SURVEY 8(d) bytes — the minimizer list is an artefact of "
                                 "this implementation, see intermediate_bytes).  Both stages are bound by VALU issue, not by HBM "
                                 "(roofline_valu, roofline_valu_jump): the fraction of the HBM peak is small by construction"},
            "roofline_valu": valu_roofline("k_minimizer_fast", k1_avg_s),
            "roofline_valu_jump": valu_roofline("k_jump_bin", kj_avg_s),
            "k_jump_bin": {"launches": int(n_kj), "avg_launch_us": kj_avg_s * 1e6,
                           "note": "jump hash of the minimizer list, alone (the `kernels` leg)"},
            "k_jump_left": {"launches": int(n_kl), "avg_launch_us": kl_avg_s * 1e6,
                            "note": "the chains k_jump_bin handed over (at most 10 lanes of a round still running)"},
            "ms_per_step_kernels_alone": instr_pass["elapsed"] / instr_pass["steps"] * 1e3,
            "roofline_cws_scan": {"bound": "hbm", "kernel": "k_cws_scan", "achieved": achieved,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "traffic": None, "traffic_from_profile": from_profile("k_cws_scan_list", "hbm_bytes_per_launch") or from_profile("k_cws_scan", "hbm_bytes_per_launch"),
                                  "launches": int(n_launch),
                                  "avg_launch_us": avg_s * 1e6, "alg_bytes_per_launch": alg_bytes,
                                  "intervals_per_launch": BATCH, "tiles_read_per_launch": visited,
                                  "tiles_covered_per_launch": covered, "unpruned_bytes_per_launch": full_bytes,
                                  "note": "exact branch-and-bound: a batch whose smallest count-min counter already rules "
                                          "out every slot skips the estimates/scan/resolve, otherwise only tiles whose "
                                          "lower bound can beat a slot's current weight are read (identical sketch; "
                                          "--no-prune / HULK_FLAG_NO_PRUNE disables both = value_unpruned)"},
        })
        if rank == 0:
            assert np.array_equal(instr_pass["mins"], mins) and np.array_equal(instr_pass["weights"], weights), \
                "the one-stream pass computed another sketch than the timed one"

    if instr_pass is not None and rank == 0:
        run_leg("rooflines", add_rooflines)

    single = args.single_pass or args.no_prune
    plain_single = world == 1 and not use_dist and not loop_world
    # ---- leg `unpruned`: the same K steps with the exact pruning of the CWS scan switched off: every interval streams the
    # whole table, as the reference's algorithm does (sketch asserted identical)
    if not single:
        def unpruned():
            full = run_pass(False, brackets=0)
            if rank == 0:
                assert np.array_equal(full["mins"], mins) and np.array_equal(full["weights"], weights), "pruning changed the sketch"
            out["value_unpruned"] = total_reads / full["elapsed"]
            out["ms_per_step_unpruned"] = full["elapsed"] / steps * 1e3
        run_leg("unpruned", unpruned, collective=True)
    elif args.no_prune:
        out["value_unpruned"] = value
    # ---- leg `scan_unpruned`: the one HBM-bound kernel of the path priced on a FULL pass: HULK_FLAG_NO_PRUNE + NO_OVERLAP
    # (k_cws_scan alone, bracketed by HIP events on its stream): 4 * slots * k^4 bytes of K32 + the two reciprocal vectors
    # per launch over its average duration (profiles/r06_kernel_stats_serial_noprune.md holds the same figure from rocprofv3)
    if not single and plain_single and rank == 0:
        def scan_unpruned():
            p = run_pass(False, brackets=8 | 32, serial=True, n_steps=min(steps, 10))
            n_l, ms_ = p["prof"]["k_cws_scan"]
            bytes_ = 4.0 * sc * (((K ** 4 + 1023) // 1024) * 1024) + 2 * 4.0 * (K ** 4)
            avg = (ms_ / 1e3) / max(n_l, 1)
            assert np.array_equal(p["mins"], mins) and np.array_equal(p["weights"], weights)
            out["roofline_cws_scan_unpruned"] = {"bound": "hbm", "kernel": "k_cws_scan_list<2> (every tile of K32 read: HULK_FLAG_NO_PRUNE)",
                                                 "achieved": bytes_ / avg / 1e9 if avg > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": (bytes_ / avg / 1e9 / HBM_PEAK_GBS) if avg > 0 else 0.0, "traffic": None,
                                                 "launches": int(n_l), "avg_launch_us": avg * 1e6, "alg_bytes_per_launch": bytes_}
            # the whole scan STAGE of a flush — the three small kernels in front of the scan (per-tile extrema of the reciprocals, the bound
            # test that builds the tile list, the per-bin max / min vectors) and the scan itself — from the every-launch table of the same pass
            tb = p.get("table") or {}
            stage = [kk for kk in ("k_rcp_extrema", "k_scan_test", "k_rcp_minmax", "k_cws_scan_list", "k_cws_scan") if kk in tb]
            if stage and n_l:
                st_us = sum(tb[kk][1] for kk in stage) * 1e3 / n_l
                out["roofline_cws_scan_unpruned"].update({"stage_kernels": stage, "stage_us_per_flush": st_us,
                                                          "frac_whole_stage": bytes_ / (st_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                          "stage_note": "avg_launch_us / frac: the scan kernel alone (HIP events around it); stage_us_per_flush / frac_whole_stage: "
                                                                        "with the three kernels that prepare it (hulk_set_profiling(32): time to the next launch's event)"})
        run_leg("scan_unpruned", scan_unpruned)
    # ---- leg `long`: the timed pass again with >= 200 steps (the 20 steps of the headline still carry the tail of the clock
    # ramp of their own pass: --steps 20 / 100 / 400 gave 0.9946 / 0.9863 / 0.9842 ms per step in round 3)
    if not single and not args.no_long and LONG_STEPS > steps:
        def long_pass():
            lp = run_pass(not args.no_prune, brackets=0, n_steps=LONG_STEPS)
            out["ms_per_step_long"] = lp["elapsed"] / LONG_STEPS * 1e3
            out["value_long"] = LONG_STEPS * reads_per_step / lp["elapsed"]
            out["steps_long"] = LONG_STEPS
            if rank == 0:
                # (LONG_STEPS steps cycle through the same 22 distinct step buffers: another stream than the headline's, so
                #  only the counters are compared)
                assert lp["counters"]["n_reads"] == (LONG_STEPS + warmup) * reads_per_rank_step
        run_leg("long", long_pass, collective=True)
    if plain_single and not args.no_cold:
        cold = run_leg("cold", run_cold)
        if cold:
            out.update(cold)
    # N > 1: the same K steps under the other modes too, so one driver run yields all of them
    if use_dist and not single:      # (world 1 only with --force-collective: a test of this path)
        other = []
        del main_input[0][:]
        torch.cuda.empty_cache()
        for om in ("sharded", "sliced-strong", "sliced-weak"):
            if om == mode:
                continue

            def other_mode(om=om):
                oin = make_input(om, 8)
                op = run_pass(True, oin, brackets=0)
                other.append({"mode": om, "value": steps * share(om)[2] / op["elapsed"], "ms_per_step": op["elapsed"] / steps * 1e3,
                              "reads_per_rank_step": oin[3], "reads_per_step": share(om)[2],
                              "sketch_md5": hashlib.md5(op["mins"].astype("<u8").tobytes()).hexdigest(), "exchange": op["comm"]})
                del oin[0][:]
                torch.cuda.empty_cache()
            run_leg("other_scaling_" + om, other_mode, collective=True)
        out["other_scaling"] = other
        if not args.no_c4:
            c4 = run_leg("c4", run_c4, collective=True)
            if c4:
                out.update(c4)
    if plain_single and not args.no_c3:
        del main_input[0][:]
        torch.cuda.empty_cache()
        c3 = run_leg("c3", run_c3)
        if c3:
            out["c3"] = c3
    if plain_single and rank == 0 and not args.no_long_reads:
        lr = run_leg("long_reads", run_long_reads)
        if lr:
            out["long_reads"] = lr
    if plain_single and rank == 0 and not args.no_c5:
        c5 = run_leg("c5", c5_leg)
        if c5:
            out["c5"] = c5
    if plain_single and rank == 0 and not args.no_e2e:
        e2e = run_leg("e2e", e2e_file_rates)
        if e2e:
            out["e2e"] = e2e
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cb = run_leg("cpu_baseline", cpu_baseline)
        if cb:
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu"] = value / cb["value"]
            if out.get("value_cold"):
                out["speedup_vs_cpu_cold"] = out["value_cold"] / cb["value"]
    for name, err in leg_errors.items():
        out[f"{name}_error"] = err
    state["leg"], state["kick"] = "exit", time.monotonic()
    print_line()
    if use_dist:
        try:
            dist.destroy_process_group()
        except Exception:             # noqa: BLE001 — the line is out; a rank that left a collective early must not turn that into rc != 0
            pass


if __name__ == "__main__":
    main()
