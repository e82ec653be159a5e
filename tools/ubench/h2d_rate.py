#!/usr/bin/env python3
"""Pinned host memory -> device, 16 MiB copies back to back on one stream (what bounds FASTQ / FASTA file -> sketch): GB/s."""
import time, torch
n = 16 << 20
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3)]
s = torch.cuda.Stream()
for reps in (8, 160, 160):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for i in range(reps):
            d[i % 3].copy_(h[i % 4], non_blocking=True)
    s.synchronize(); dt = time.perf_counter() - t0
    print("%d x 16 MiB: %.1f GB/s" % (reps, reps * n / dt / 1e9))
