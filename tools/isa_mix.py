#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel, priced with the issue costs measured on the chip (profiles/r03_op_cost.txt,
tools/ubench/op_cost2.hip: cycles per wave64 instruction per SIMD with the SIMD saturated):
  fast   2.35  v_add/sub_u32, v_and/or/xor_b32, v_mov_b32, v_lshrrev_b32, v_mul_f32, v_add_f32 (VOP2 form; _e64 2.48)
  fmac   2.67 / v_fma_f32 3.66
  full   4.4   everything else that is one pass of the SIMD-16: v_lshlrev_b32, min/max, mul_u32_u24, add_co/addc_co, every
               VOP3 (alignbit, bfe, perm, and_or, lshl_add, add3, mad_u32_u24, mul_lo, mad_u64_u32), DPP moves, cvt, cmp,
               cndmask, every 64-bit / fp64 / packed-fp32 op, readlane/readfirstlane
  trans  8.3 v_rcp_f32-class, 16.5 v_rcp_f64-class
usage: isa_mix.py <file.s> <kernel-name-substring> [--loop]     (file.s: hipcc --cuda-device-only -S)
Prints the counts over the kernel's text and over its largest loop (label .. backward branch), and the mean cycles per
VALU instruction — the weight bench.py's roofline_valu uses instead of one figure for every instruction."""
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
        "v_mul_f32", "v_add_f32", "v_sub_f32", "v_not_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
COST = {"fast": 2.35, "fast_e64": 2.48, "fmac": 2.67, "fma32": 3.66, "full": 4.4, "trans32": 8.3, "trans64": 16.5}


def classify(mn, ops):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if "dpp" in mn or "row_" in ops or "quad_perm" in ops:
        return "full"
    if base in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"):
        return "trans64"
    if base in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"):
        return "trans32"
    if base in ("v_fmac_f32", "v_mac_f32"):
        return "fmac"
    if base == "v_fma_f32":
        return "fma32"
    if base in FAST:
        return "fast_e64" if mn.endswith("_e64") else "fast"
    return "full"


def mix(lines):
    c = {}
    for mn, ops in lines:
        k = classify(mn, ops)
        c[k] = c.get(k, 0) + 1
    n = sum(c.values())
    cyc = sum(COST[k] * v for k, v in c.items())
    return c, n, (cyc / n if n else 0.0)


def kernel_body(path, needle):
    out, on = [], False
    for line in open(path):
        if not on:
            m = re.match(r"^(\S+):\s*(;.*)?$", line)
            if m and needle in m.group(1) and not m.group(1).startswith("."):
                on = True
            continue
        if line.startswith("\t.end_amdhsa_kernel") or re.match(r"^\s*\.size\s", line) or line.startswith(".Lfunc_end"):
            break
        out.append(line.rstrip("\n"))
    return out


def analyse(path, needle):
    """(counts, n VALU, mean cycles) over the kernel's text and over its largest loop"""
    body = kernel_body(path, needle)
    if not body:
        return None
    labels, insts = {}, []
    for line in body:
        m = re.match(r"^(\.L\w+):", line)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(;.*)?$", line)
        if m and not m.group(1).startswith("."):
            insts.append((m.group(1), m.group(2)))
    whole = mix([(mn, ops) for mn, ops in insts if mn.startswith("v_")])
    best = None
    for i, (mn, ops) in enumerate(insts):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            tgt = ops.strip().split()[0] if ops.strip() else ""
            if tgt in labels and labels[tgt] <= i:
                seg = [(a, b) for a, b in insts[labels[tgt]:i + 1] if a.startswith("v_")]
                if best is None or len(seg) > len(best):
                    best = seg
    return whole, (mix(best) if best else None)


def main():
    if sys.argv[1] == "--json":          # isa_mix.py --json out.json name=file.s:needle ...
        import json
        out = {"costs_cycles": COST, "source": "profiles/r03_op_cost.txt (tools/ubench/op_cost2.hip)",
               "note": "static mix of the production instance's text / of its largest loop; mean = sum(count x cost) / count"}
        for spec in sys.argv[3:]:
            name, rest = spec.split("=", 1)
            path, needle = rest.split(":", 1)
            r = analyse(path, needle)
            if r is None:
                sys.exit(f"no kernel matching {needle!r} in {path}")
            whole, loop = r
            out[name] = {"symbol_contains": needle, "valu_in_text": whole[1], "mix_text": whole[0], "mean_cycles_text": whole[2],
                         "valu_in_loop": loop[1] if loop else None, "mix_loop": loop[0] if loop else None,
                         "mean_cycles_loop": loop[2] if loop else None}
        json.dump(out, open(sys.argv[2], "w"), indent=1)
        return
    path, needle = sys.argv[1], sys.argv[2]
    body = kernel_body(path, needle)
    if not body:
        sys.exit(f"no kernel matching {needle!r} in {path}")
    labels, insts = {}, []           # label -> index into insts; insts: (mnemonic, operands)
    for line in body:
        m = re.match(r"^(\.L\w+):", line)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(;.*)?$", line)
        if m and not m.group(1).startswith("."):
            insts.append((m.group(1), m.group(2)))
    valu = [(mn, ops) for mn, ops in insts if mn.startswith("v_")]
    c, n, mean = mix(valu)
    print(f"{needle}: {len(insts)} instructions in the text, {n} VALU: {c}; mean {mean:.2f} cycles per VALU instruction")
    best = None
    for i, (mn, ops) in enumerate(insts):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            tgt = ops.strip().split()[0] if ops.strip() else ""
            if tgt in labels and labels[tgt] <= i:
                seg = [(a, b) for a, b in insts[labels[tgt]:i + 1] if a.startswith("v_")]
                if best is None or len(seg) > len(best[0]):
                    best = (seg, tgt, i - labels[tgt] + 1)
    if best:
        c, n, mean = mix(best[0])
        print(f"  largest loop ({best[1]}, {best[2]} instructions, {n} VALU): {c}; mean {mean:.2f} cycles per VALU instruction")


if __name__ == "__main__":
    main()
