"""Run one BASELINE config on the GPU path with synthetic HBM-resident reads and report timing.
usage: run_config.py --k 31 --S 1024 --decay 0.02 --reads 5000000 --interval 100000 [--batch 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=31); ap.add_argument("--w", type=int, default=9)
ap.add_argument("--S", type=int, default=1024); ap.add_argument("--decay", type=float, default=0.02)
ap.add_argument("--reads", type=int, default=5_000_000); ap.add_argument("--interval", type=int, default=100_000)
ap.add_argument("--batch", type=int, default=10); ap.add_argument("--len", type=int, default=150)
ap.add_argument("--lanes", type=int, default=0, help="hulk_params.work_lanes (0 = the library default, 2)")
ap.add_argument("--serial", action="store_true", help="HULK_FLAG_NO_OVERLAP: every kernel alone (profiling)")
ap.add_argument("--no-prune", action="store_true", help="HULK_FLAG_NO_PRUNE: every interval against the whole CWS table")
a = ap.parse_args()
import torch, hulk_amd
from hulk_amd import synth
t0 = time.time()
sk = hulk_amd.GpuSketcher(a.k, a.w, a.S, interval=a.interval, decay_ratio=a.decay,
                          batch=a.batch, work_lanes=a.lanes,
                          flags=(16 if a.serial else 0) | (2 if a.no_prune else 0))
torch.cuda.synchronize(); t_create = time.time() - t0
step = a.interval * a.batch
bufs = []
for s in range(min(4, (a.reads + step - 1) // step)):
    b, o = synth.reads_torch(s * step, step, a.len); bufs.append((b, o))
torch.cuda.synchronize()
done = 0; i = 0
sk.add_reads_device(bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), step, a.len, bufs[0][0].numel()); done += step; i += 1   # warm
sk.synchronize()                       # (the context runs on its private streams: its own synchronisation points bracket the clock)
t1 = time.perf_counter()
while done < a.reads:
    b, o = bufs[i % len(bufs)]
    sk.add_reads_device(b.data_ptr(), o.data_ptr(), step, a.len, b.numel()); done += step; i += 1
sk.synchronize()
ms = (time.perf_counter() - t1) * 1e3
tiles = sk.scan_stats()
sk.finish()
mins, w = sk.sketch()
print(json.dumps({"k": a.k, "S": a.S, "decay": a.decay, "interval": a.interval, "batch": a.batch,
                  "reads_timed": done - step, "ms": ms, "reads_per_s": (done - step) / ms * 1e3,
                  "create_s": t_create, "mem_GB": torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9,
                  "scan_tiles_read": tiles[0], "scan_tiles_covered": tiles[1], "distinct_mins": int(len(set(mins.tolist()))), "neg_weights": int((w < 0).sum())}))
