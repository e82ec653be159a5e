#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the markdown summary committed under profiles/.
usage: rocprof_summary.py <results.db> <out.md> "<title>" "<command>" """
import re
import sqlite3
import sys

db, out, title, cmd = sys.argv[1:5]
c = sqlite3.connect(db)
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(out, "w") as f:
    f.write(f"# {title}\n\nCommand: `{cmd}` (MI355X, 1 GPU, rocprofv3 --kernel-trace --stats).\n")
    f.write("Durations in microseconds (`top_kernels` view of the rocpd database). `at::native` kernels are\n"
            "torch's synthetic-read generator / collectives plumbing, outside the timed region.\n\n")
    f.write("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|\n")
    for n, cl, t, a, p in rows:
        m = re.search(r"(k_\w+)[<(]", n)
        short = m.group(1) if m else re.sub(r"[|<].*", "", n.replace("void ", ""))[:70]
        f.write(f"| {short} | {cl} | {t:.1f} | {a:.2f} | {p:.2f} |\n")
print(open(out).read())
