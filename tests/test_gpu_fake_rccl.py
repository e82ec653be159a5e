"""The library's RCCL branch (hulk_comm.hip, transport kind 1: grouped in-place ncclAllGather of payload + header per sharded
step, the in-place uint32 ncclAllReduce of hulk_step_sliced, the EOF all-gather of the slot shards) WITH PEERS: 2, 3 and 8
processes on the one GPU of this box, each a product rank that calls hulk_comm_init (ncclCommInitRank underneath).

RCCL itself refuses a communicator whose ranks share a device (profiles/r05_rccl_world2.txt), so the nccl* symbols are bound
to tests/cpp/fake_rccl.cpp (HULK_RCCL_LIB): a test double over POSIX shared memory that keeps what the product relies on —
asynchronous and stream-ordered collectives, group semantics, in-place buffers — and is stricter than the real thing about
one: every operation carries {sequence number, kind, bytes}, and ranks that disagree (one took the delta exchange, its peer the
spectra exchange) abort with both descriptors instead of hanging.  Nothing else is swapped: which exchange a step takes, the
kernels and their order are the ones real RCCL ranks run.  The gathered sketch must be the single-rank sketch of the same
global stream (SeqMinimizer.Run's interval rule, src/pipeline/sketch.go:211-224), and the oracle's."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "cpp", "libfakerccl.so")
K, W, S, I, BATCH, L = 15, 9, 64, 3000, 4, 150
STEPS, TAIL = 3, 2                            # sliced mode: whole batches + a tail batch of TAIL intervals


def _fake_lib():
    """built by __graft_entry__.build(); (re)built here if the snapshot lacks it"""
    src = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", FAKE, src, "-lrt", "-lpthread"], check=True)
    return FAKE


def _total(world):
    return 4 * world * BATCH * I + I + 1100   # four whole steps + a ragged one (rank 0: a whole and a partial interval, the others nothing)


def _worker(rank, world, uid_q, out_q, mode, inject):
    import ctypes
    import torch
    import hulk_amd
    from hulk_amd import _lib, synth
    from hulk_amd.distributed import interval_slice, num_steps, slot_shard, step_share
    torch.cuda.set_device(0)
    if rank == 0:
        uid = hulk_amd.GpuSketcher.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=900)      # (the first `import torch` of a fresh box takes minutes: rank 0 may be late)
    sb, sc = slot_shard(S, rank, world)
    sharded = mode.startswith("sharded")
    flags = _lib.HULK_FLAG_SHARD_FULL if mode == "sharded-full" else 0
    sk = hulk_amd.GpuSketcher(K, W, S, interval=I if sharded else 0, decay_ratio=1.0, device=0, slot_begin=sb, slot_count=sc,
                              batch=BATCH, flags=flags | _lib.HULK_FLAG_NO_PRERESERVE)
    sk.comm_init(uid, rank, world)                              # ncclCommInitRank of the library bound through HULK_RCCL_LIB
    if inject is not None and inject[0] == rank:
        sk.debug_inject(inject[1], inject[2])
    keep, err = [], None
    total = _total(world)
    try:
        if sharded:
            for s_ in range(num_steps(total, BATCH, I, world)):
                first, n, step_intervals = step_share(s_, BATCH, I, rank, world, total)
                b, off = synth.reads_torch(first, max(n, 1), L, device="cuda:0")
                keep.append((b, off))
                torch.cuda.synchronize()
                sk.step_sharded(b.data_ptr(), off.data_ptr(), n, L, b.numel(), step_intervals)
        else:
            per = interval_slice("strong", 0, I, rank, world)[1]
            for s_ in range(STEPS + 1):
                nt = BATCH if s_ < STEPS else TAIL
                parts, cnts = [], []
                for t in range(nt):
                    first, cnt = interval_slice("strong", s_ * BATCH + t, I, rank, world)
                    b, _ = synth.reads_torch(first, cnt, L, device="cuda:0")
                    parts.append(b[:cnt * L]); cnts.append(cnt)
                assert len(set(cnts)) == 1 and cnts[0] == per     # (I divisible by the world sizes used: equal slices)
                bases = torch.cat(parts + [torch.zeros(16, dtype=torch.uint8, device="cuda:0")])
                offsets = torch.arange(per * nt + 1, dtype=torch.int64, device="cuda:0") * L
                keep.append((bases, offsets))
                torch.cuda.synchronize()
                sk.step_sliced(bases.data_ptr(), offsets.data_ptr(), per * nt, L, bases.numel(), per, nt)
        sk.finish()
        mins, weights = sk.gather_sketch()
        cms = sk.cms()
    except hulk_amd.HulkError as e:
        err, mins, weights, cms = str(e), None, None, None
    stats = sk.comm_stats() if err is None else None
    fk = ctypes.CDLL(os.environ["HULK_RCCL_LIB"])                # the same handle the library bound: how much went through the double
    a, b_, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    fk.fakeRcclStats(ctypes.byref(a), ctypes.byref(b_), ctypes.byref(c))
    out_q.put((rank, mins, weights, cms, stats, err, (a.value, b_.value, c.value)))
    sk.close()


def _run_ranks(world, mode, inject=None, exp=False, sync=False):
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    saved = {k: os.environ.get(k) for k in ("HULK_RCCL_LIB", "HULK_LIB", "FAKE_RCCL_SYNC", "FAKE_RCCL_TIMEOUT_S")}
    os.environ["HULK_RCCL_LIB"] = _fake_lib()
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "600"              # (a peer that is still importing torch on a cold box is not a hang)
    if exp:
        os.environ["HULK_LIB"] = "exp"
    if sync:
        os.environ["FAKE_RCCL_SYNC"] = "1"
    try:
        ctx = mp.get_context("spawn")
        uid_q, out_q = ctx.Queue(), ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, uid_q, out_q, mode, inject)) for r in range(world)]
        for p in procs:
            p.start()
        res = {}
        try:
            import queue
            import time
            t_end = time.time() + 1500
            while len(res) < world and time.time() < t_end:
                try:
                    r = out_q.get(timeout=5)
                    res[r[0]] = r
                except queue.Empty:
                    if any(p.exitcode not in (None, 0) for p in procs):      # a rank died (the double aborts on a protocol error): no point in waiting
                        break
        finally:
            for p in procs:
                p.join(timeout=60)
                if p.is_alive():
                    p.kill()                                     # (exact processes this test started)
        assert len(res) == world and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        return [res[r] for r in range(world)]
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


_single = {}


def _single_rank(total):
    import hulk_amd
    from hulk_amd import synth
    from oracle import pyorc
    if total not in _single:
        bases, offsets = synth.reads_numpy(0, total, L)
        g = hulk_amd.GpuSketcher(K, W, S, interval=I)
        g.add_reads(bases, offsets)
        g.finish()
        m, w = g.sketch()
        c = g.cms()
        g.close()
        o = pyorc.Sketcher(K, W, S, 0, 1.0, I)
        o.add_reads(bases, offsets)
        o.finish()
        mo, wo = o.sketch()
        assert np.array_equal(m, mo) and np.allclose(w, wo, rtol=1e-12, atol=0) and np.array_equal(c, o.cms())
        o.close()
        _single[total] = (m, w, c)
    return _single[total]


@pytest.mark.parametrize("world,mode", [(2, "sharded"), (3, "sharded"), (8, "sharded"), (2, "sharded-full"),
                                        (2, "sliced-strong"), (3, "sliced-strong"), (8, "sliced-strong")])
def test_rccl_branch_with_peers_gives_the_single_rank_sketch(world, mode):
    out = _run_ranks(world, mode)
    total = _total(world) if mode.startswith("sharded") else (STEPS * BATCH + TAIL) * I
    m1, w1, c1 = _single_rank(total)
    for rank, mins, weights, cms, stats, err, fk in out:
        assert err is None, err
        assert np.array_equal(mins, m1) and np.array_equal(weights, w1) and np.array_equal(cms, c1), rank
        assert stats["headers_refetched"] == 0 and stats["void_blocks"] == 0, stats      # hulk_get_comm_health
        assert fk[0] == 1 and fk[1] >= 2, fk                       # one communicator of the double, its collectives ran
        if mode == "sharded":
            assert stats["steps_full"] >= 1 and stats["steps_delta"] >= 1, stats          # both exchanges were taken
            assert fk[1] == 2 * (stats["steps_full"] + stats["steps_delta"]) + 1, (fk, stats)   # payload + header per step, the EOF gather
        elif mode == "sharded-full":
            assert stats["steps_delta"] == 0
        else:
            assert fk[1] == STEPS + 1 + 1                           # one all-reduce per step, the EOF gather


def test_rccl_branch_blocking_double_agrees():
    """the same run with the double's collectives blocking the host (FAKE_RCCL_SYNC): the product does not depend on the
    collective returning before it has run"""
    out = _run_ranks(2, "sharded", sync=True)
    m1, w1, c1 = _single_rank(_total(2))
    for rank, mins, weights, cms, stats, err, fk in out:
        assert err is None and np.array_equal(mins, m1) and np.array_equal(weights, w1) and np.array_equal(cms, c1)


def test_void_header_on_the_rccl_branch_is_loud_on_every_rank():
    """hulk_debug_inject (profiling build) on the RCCL branch: a block sealed with another step's tag
      * in a spectra-exchange step: every rank takes the spectra exchange once more, the sketch is the single-rank sketch;
      * in a delta step: HULK_ERR_COMM on EVERY rank, and every rank leaves its collectives (none hangs: the double would
        report a missing peer)."""
    from hulk_amd import _lib
    if not os.path.exists(os.path.join(ROOT, "hulk_amd", "csrc", "libhulkhip_exp.so")):
        pytest.skip("profiling build (make -C hulk_amd/csrc EXPERIMENTS=1) not present")
    world = 3
    base = _run_ranks(world, "sharded", exp=True)
    nf, nd = base[0][4]["steps_full"], base[0][4]["steps_delta"]
    assert nf >= 1 and nd >= 2, base[0][4]
    m1, w1, c1 = _single_rank(_total(world))
    r1 = _run_ranks(world, "sharded", inject=(1, _lib.HULK_INJECT_STALE_SEAL, nf - 1), exp=True)
    for rank, mins, weights, cms, stats, err, fk in r1:
        assert err is None, err
        assert stats["steps_full"] == nf + 1 and stats["void_blocks"] == 1, stats
        assert np.array_equal(mins, m1) and np.array_equal(weights, w1) and np.array_equal(cms, c1)
    r2 = _run_ranks(world, "sharded", inject=(2, _lib.HULK_INJECT_STALE_SEAL, nf + 1), exp=True)
    for rank, mins, weights, cms, stats, err, fk in r2:
        assert err is not None and "exchange between the ranks failed" in err, err
