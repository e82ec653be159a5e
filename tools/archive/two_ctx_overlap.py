#!/usr/bin/env python3
"""Experiment: do the VALU-bound minimizer and jump kernels fill each other's issue bubbles?  Two independent contexts
(private streams) are fed alternately with the bench's step; aggregate reads/s vs one context alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hulk_amd
from hulk_amd import synth
K, W, S, L, I, T = 21, 9, 512, 150, 100_000, 16
dev = torch.device("cuda:0")
n = I * T
bufs = []
for s_ in range(4):
    b, _ = synth.reads_torch(s_ * n, n, L, device=dev)
    bufs.append(torch.cat([b[:n * L], torch.zeros(16, dtype=torch.uint8, device=dev)]))
off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
torch.cuda.synchronize()
def run(nctx, steps=24):
    sks = [hulk_amd.GpuSketcher(K, W, S, interval=0, decay_ratio=1.0, batch=T, work_lanes=int(os.environ.get('LANES', '1'))) for _ in range(nctx)]
    def step(t):
        for sk in sks:
            b = bufs[t % 4]
            sk.bin_reads_device(b.data_ptr(), off.data_ptr(), n, L, b.numel(), reads_per_spectrum=I)
            sk.flush_batch(T)
    for t in range(3): step(t)
    for sk in sks: sk.counters()
    t0 = time.perf_counter()
    for t in range(steps): step(t)
    for sk in sks: sk.counters()
    dt = time.perf_counter() - t0
    for sk in sks: sk.close()
    return nctx * steps * n / dt
r1 = run(1); r2 = run(2); r3 = run(3)
print(f"1 context {r1:.3e} reads/s   2 contexts {r2:.3e} ({r2 / r1:.3f}x)   3 contexts {r3:.3e} ({r3 / r1:.3f}x)")
