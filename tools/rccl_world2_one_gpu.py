#!/usr/bin/env python3
"""Can RCCL build a 2-rank communicator when both ranks sit on the ONE GPU of this box?  Two processes, both on device 0,
call hulk_comm_init (ncclCommInitRank underneath) with world = 2 and, if that succeeds, run a few hulk_step_sharded steps
and compare the gathered sketch with a single-rank run.  Prints what RCCL says either way.
usage: rccl_world2_one_gpu.py            (run on the GPU box; ends by itself within ~2 minutes)"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K, W, S, I, BATCH, L = 15, 9, 64, 3000, 4, 150
WORLD = 2
TOTAL = 3 * WORLD * BATCH * I + I + 1100


def worker(rank, uid_q, out_q):
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    import torch
    import hulk_amd
    from hulk_amd import synth
    from hulk_amd.distributed import num_steps, slot_shard, step_share
    torch.cuda.set_device(0)
    if rank == 0:
        uid = hulk_amd.GpuSketcher.comm_unique_id()
        uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=60)
    sb, sc = slot_shard(S, rank, WORLD)
    sk = hulk_amd.GpuSketcher(K, W, S, interval=I, device=0, slot_begin=sb, slot_count=sc, batch=BATCH)
    t0 = time.time()
    try:
        sk.comm_init(uid, rank, WORLD)
    except hulk_amd.HulkError as e:
        out_q.put((rank, "comm_init refused", str(e), time.time() - t0))
        return
    keep = []
    for s_ in range(num_steps(TOTAL, BATCH, I, WORLD)):
        first, n, si = step_share(s_, BATCH, I, rank, WORLD, TOTAL)
        b, off = synth.reads_torch(first, max(n, 1), L, device="cuda:0")
        keep.append((b, off))
        torch.cuda.synchronize()
        sk.step_sharded(b.data_ptr(), off.data_ptr(), n, L, b.numel(), si)
    sk.finish()
    m, w = sk.gather_sketch()
    out_q.put((rank, "ran", (m.tolist(), w.tolist(), sk.comm_stats()), time.time() - t0))
    sk.close()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    uid_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, uid_q, out_q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = []
    deadline = time.time() + 100
    while len(res) < WORLD and time.time() < deadline:
        try:
            res.append(out_q.get(timeout=2))
        except Exception:  # noqa: BLE001
            if not any(p.is_alive() for p in procs):
                break
    for p in procs:
        p.join(timeout=5)
        if p.is_alive():
            p.kill()                                              # (exact PIDs this script started)
    if len(res) < WORLD:
        print(f"RCCL world 2 on one GPU: {len(res)} of {WORLD} ranks reported within 100 s (hung in ncclCommInitRank or a collective): {res and res[0][:3]}")
        sys.exit(0)
    res.sort()
    if all(r[1] == "ran" for r in res):
        import numpy as np
        import hulk_amd
        from hulk_amd import synth
        bases, offsets = synth.reads_numpy(0, TOTAL, L)
        g = hulk_amd.GpuSketcher(K, W, S, interval=I)
        g.add_reads(bases, offsets); g.finish()
        m1, w1 = g.sketch(); g.close()
        same = all(np.array_equal(np.array(r[2][0], dtype=np.uint64), m1) and np.array_equal(np.array(r[2][1]), w1) for r in res)
        print(f"RCCL world 2 on one GPU: RAN ({res[0][3]:.1f} s); gathered sketch == single-rank sketch: {same}; stats {res[0][2][2]}")
    else:
        for r in res:
            print(f"RCCL world 2 on one GPU: rank {r[0]}: {r[1]}: {r[2]} ({r[3]:.1f} s)")
