"""BASELINE config C5: pairwise weighted-Jaccard matrix over 1024 sketches (sketchSize=2048) on 1 GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hulk_amd.smash import distance_matrix
from oracle import pyorc
rng = np.random.default_rng(5)
N, S = 1024, 2048
base = rng.integers(0, 194481, size=S).astype(np.uint64)
mins = np.where(rng.random((N, S)) < 0.5, base, rng.integers(0, 194481, size=(N, S)).astype(np.uint64))
w = -rng.gamma(2.0, 1e-3, size=(N, S))
distance_matrix(mins[:8], w[:8], "weightedjaccard")
for metric in ("weightedjaccard", "jaccard"):
    t0 = time.perf_counter(); d = distance_matrix(mins, w, metric); dt = time.perf_counter() - t0
    print(f"GPU {metric}: {dt * 1e3:.1f} ms end to end (H2D 32 MB + kernel + D2H 8 MB) = {N * N / dt:.3e} pairs/s")
t0 = time.perf_counter(); ref = pyorc.smash_matrix(mins[:256], w[:256], "weightedjaccard"); dt = time.perf_counter() - t0
print(f"CPU oracle (1 thread), 256x256 sample: {dt:.2f} s = {256 * 256 / dt:.3e} pairs/s; identical: {np.array_equal(ref, distance_matrix(mins[:256], w[:256], 'weightedjaccard'))}")
