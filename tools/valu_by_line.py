#!/usr/bin/env python3
"""VALU instructions of ONE kernel instance by SOURCE LINE (static, from hipcc's own line table).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --cuda-device-only -o hm.s hulk_minimizer.hip
  tools/valu_by_line.py hm.s <mangled-name substring> <source.hip> [--loop] [--min N] [--md]

Every instruction of the kernel's text is attributed to the source line of the `.loc` directive in front of it (inlined callees: the
line of the callee's statement; `--callsite` is not reconstructed — device functions are listed under their own lines), counted as
VALU / SALU / LDS / VMEM, and priced with the issue costs measured on the chip (tools/isa_mix.py: cycles per wave64 instruction).
--loop restricts the table to the kernel's main loop (the largest label .. backward-branch span).  Per READ figures: the main loop
of k_minimizer_fast handles 4 reads per trip (2 with PAIR), so per-read = per-trip / 4 — printed when --reads-per-trip is given.
Static counts: a branch not taken (the N variant, the rare per-byte path of the staging) is counted at its full length, so the
table's total exceeds the PMC's dynamic count; --skip-lines a-b,c-d leaves such line ranges out."""
import argparse
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isa_mix import COST, classify, kernel_body   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm"); ap.add_argument("kernel"); ap.add_argument("source")
    ap.add_argument("--loop", action="store_true"); ap.add_argument("--min", type=int, default=1)
    ap.add_argument("--reads-per-trip", type=float, default=0.0)
    ap.add_argument("--skip-lines", default="")
    ap.add_argument("--lines", default="", help="a-b: only instructions attributed to these source lines (e.g. the body of the main loop); "
                    "instructions of inlined helpers from other files count under the last line of the source file seen in front of them (their call site, roughly)")
    ap.add_argument("--title", default="")
    ap.add_argument("--helpers-below", type=int, default=0, help="lines of the source file below this one hold inlined device helpers: counted under their call site too (with --lines)")
    a = ap.parse_args()
    body = kernel_body(a.asm, a.kernel)
    if not body:
        sys.exit(f"kernel with '{a.kernel}' in its name not found in {a.asm}")
    # file numbers of the line table
    files = {}
    for line in open(a.asm):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
    want = {n for n, f in files.items() if f.endswith(a.source.split("/")[-1])}
    src = open(a.source).read().split("\n")
    skip = set()
    for part in filter(None, a.skip_lines.split(",")):
        lo, _, hi = part.partition("-")
        skip.update(range(int(lo), int(hi or lo) + 1))
    # instruction stream with (file, line)
    ins, cur, labels = [], (0, 0), {}
    ctx = 0
    for line in body:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            if cur[0] in want and cur[1] >= a.helpers_below:
                ctx = cur[1]
            elif a.lines and ctx:
                cur = (min(want), ctx)                       # an inlined helper: under its call site
            continue
        m = re.match(r"^(\.L[\w$]+):", line)
        if m:
            labels[m.group(1)] = len(ins); continue
        m = re.match(r"^\s+([vsdgb][a-z0-9_]+)\s*(.*?)\s*(;.*)?$", line)
        if m and not m.group(1).startswith("s_nop") and not line.strip().startswith("."):
            ins.append((m.group(1), m.group(2), cur))
    lo, hi = 0, len(ins)
    if a.loop:
        best = (0, 0, 0)
        for i, (mn, ops, _) in enumerate(ins):
            if mn.startswith("s_cbranch") or mn == "s_branch":
                t = labels.get(ops.strip())
                if t is not None and t < i and i - t > best[0]:
                    best = (i - t, t, i + 1)
        lo, hi = best[1], best[2]
    keep = None
    if a.lines:
        lo_l, _, hi_l = a.lines.partition("-")
        keep = (int(lo_l), int(hi_l))
    rows = {}
    tot = {"valu": 0, "cyc": 0.0, "salu": 0, "lds": 0, "vmem": 0}
    for mn, ops, (f, ln) in ins[lo:hi]:
        key = ln if f in want else -f
        if key in skip or (keep and not (keep[0] <= key <= keep[1])):
            continue
        r = rows.setdefault(key, {"valu": 0, "cyc": 0.0, "salu": 0, "lds": 0, "vmem": 0})
        if mn.startswith("v_"):
            r["valu"] += 1; c = COST[classify(mn, ops)]; r["cyc"] += c; tot["valu"] += 1; tot["cyc"] += c
        elif mn.startswith("s_"):
            r["salu"] += 1; tot["salu"] += 1
        elif mn.startswith("ds_"):
            r["lds"] += 1; tot["lds"] += 1
        else:
            r["vmem"] += 1; tot["vmem"] += 1
    rpt = a.reads_per_trip
    print(f"# {a.title or a.kernel}\n")
    print(f"`{a.kernel}` — {'main loop' if a.loop else 'whole kernel'}: **{tot['valu']} VALU** wave-instructions ({tot['cyc']:.0f} issue cycles at the "
          f"measured costs, mean {tot['cyc'] / max(tot['valu'], 1):.2f}), {tot['salu']} SALU, {tot['lds']} LDS, {tot['vmem']} VMEM"
          + (f"; per read (/{rpt:g}): **{tot['valu'] / rpt:.1f} VALU**, {tot['cyc'] / rpt:.0f} cycles" if rpt else "") + "\n")
    print("| line | VALU | cycles | " + ("VALU / read | " if rpt else "") + "SALU | LDS | VMEM | source |")
    print("|---:|---:|---:|" + ("---:|" if rpt else "") + "---:|---:|---:|---|")
    for key in sorted(rows, key=lambda k: (k < 0, k)):
        r = rows[key]
        if r["valu"] + r["lds"] + r["vmem"] < a.min:
            continue
        text = (src[key - 1].strip() if 0 < key <= len(src) else f"(file {files.get(-key, '?')})")[:150].replace("|", "\\|")
        print(f"| {key if key > 0 else ''} | {r['valu']} | {r['cyc']:.0f} | " + (f"{r['valu'] / rpt:.1f} | " if rpt else "")
              + f"{r['salu']} | {r['lds']} | {r['vmem']} | `{text}` |")


if __name__ == "__main__":
    main()
