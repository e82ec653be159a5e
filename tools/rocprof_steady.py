#!/usr/bin/env python3
"""Steady-state per-kernel table from a rocprofv3 kernel trace (CSV): every dispatch that starts before the N-th launch of an
anchor kernel is left out — the first batch of a stream evaluates the whole CWS table (cold), which made the averages of
profiles/r05_c3_kernel_stats_serial.md meaningless for the flush kernels.
usage: rocprof_steady.py <..._kernel_trace.csv> <out.md> "<title>" "<command>" [anchor=k_minimizer_fast] [nth=2]"""
import csv
import re
import sys

path, out, title, cmd = sys.argv[1:5]
anchor = sys.argv[5] if len(sys.argv) > 5 else "k_minimizer_fast"
nth = int(sys.argv[6]) if len(sys.argv) > 6 else 2
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name") or r.get("kernel_name") or ""
        s, e = int(r.get("Start_Timestamp") or r.get("start_timestamp")), int(r.get("End_Timestamp") or r.get("end_timestamp"))
        m = re.search(r"(k_\w+)", name)
        rows.append((s, e, m.group(1) if m else re.sub(r"[|<(].*", "", name.replace("void ", ""))[:60]))
rows.sort()
starts = [s for s, e, n in rows if n == anchor]
if len(starts) < nth:
    sys.exit(f"only {len(starts)} launches of {anchor}")
t0 = starts[nth - 1]
n_batches = len(starts) - (nth - 1)
agg = {}
for s, e, n in rows:
    if s < t0 or not n.startswith("k_"):
        continue
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
with open(out, "w") as f:
    f.write(f"# {title}\n\nCommand: `{cmd}` (MI355X, 1 GPU, rocprofv3 --kernel-trace).  STEADY STATE: every dispatch in front of launch {nth} of "
            f"`{anchor}` is left out (the first batch of a stream is cold: it evaluates the whole CWS table) — {n_batches} batches remain.  "
            f"Durations in microseconds.\n\n")
    f.write("| kernel | calls | calls / batch | total_us | avg_us | us / batch | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"| {n} | {c} | {c / n_batches:.2f} | {t / 1e3:.1f} | {t / 1e3 / c:.2f} | {t / 1e3 / n_batches:.1f} | {100.0 * t / tot:.2f} |\n")
    f.write(f"\nSum over the kernels: {tot / 1e3 / n_batches:.1f} us per batch.\n")
print(open(out).read())
