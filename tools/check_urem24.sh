#!/bin/bash
# Does any kernel of libhulkhip.so contain hipcc's fp32 expansion of a 24-bit integer division / remainder (wrong for ~0.1 % of the
# operand pairs on gfx950: tools/ubench/urem24_check.hip)?  Its signature is the one-sided correction `v_cmp_ge_f32 |r|, d`.
# Compiles every .hip of hulk_amd/csrc to device assembly (no GPU needed) and counts.  -> profiles/r06_urem24.txt
cd "$(dirname "$0")/../hulk_amd/csrc" || exit 1
T=$(mktemp -d); n=0
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o $T/${f%.hip}.s $f 2>/dev/null & done; wait
for s in $T/*.s; do c=$(grep -c 'v_cmp_ge_f32_e64.*|' $s); n=$((n + c)); echo "$(basename $s .s): $c"; done
echo "24-bit division expansions in the library's device code: $n"
rm -rf $T; [ $n -eq 0 ]
