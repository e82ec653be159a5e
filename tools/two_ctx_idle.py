#!/usr/bin/env python3
"""Diagnosis (round 4): is a context slower while a SECOND context of the process is alive (idle)?  Found with bench.py's
first clock-ramp variant (a throw-away context): 1.20 instead of 0.96 ms per step with the lanes' buffers reserved by
hulk_create.  Prints ms per step of context A alone, with an idle B created after / before it, and after B is closed."""
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
def make(): return hulk_amd.GpuSketcher(K, W, S, interval=I, decay_ratio=1.0, batch=T)
def run(sk, steps=60, warm=8):
    for t in range(warm): sk.add_reads_device(bufs[t % 4].data_ptr(), off.data_ptr(), n, L, bufs[t % 4].numel())
    sk.synchronize(); t0 = time.perf_counter()
    for t in range(steps): sk.add_reads_device(bufs[t % 4].data_ptr(), off.data_ptr(), n, L, bufs[t % 4].numel())
    sk.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
def mem(): f, t = torch.cuda.mem_get_info(); return (t - f) / 1e9
a = make(); run(a, 40)
print("A alone: %.4f %.4f ms/step, %.1f GB in use" % (run(a), run(a), mem()))
b = make()
print("A with idle B (created after A): %.4f %.4f, %.1f GB" % (run(a), run(a), mem()))
print("B with idle A: %.4f %.4f" % (run(b), run(b)))
print("A again: %.4f %.4f" % (run(a), run(a)))
b.close()
print("A after B closed: %.4f %.4f, %.1f GB" % (run(a), run(a), mem()))
a.close()
c = make(); d = make()
print("fresh C (D created right after it, idle): %.4f %.4f" % (run(c), run(c)))
print("D: %.4f %.4f" % (run(d), run(d)))
d.close(); c.close()
e = make()
print("fresh E alone after all that: %.4f %.4f" % (run(e), run(e)))
