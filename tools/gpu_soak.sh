#!/bin/bash
# Fuzz soak on the GPU box: parity (default / big k / every w), the multi-rank protocol (delta and forced-full), the device parsers (FASTQ, FASTA) — each leg stops itself after the given minutes and prints its count.
# usage (through gpurun): bash tools/gpu_soak.sh SEED0 [minutes per leg, default 4]
S=${1:-9000}; M=${2:-4}; T=$((M * 60))
mkdir -p gpurun_out/soak; O=gpurun_out/soak/soak_$S.txt; : > $O
leg() { echo "== $*" >> $O; ( FUZZ_SECONDS=$T timeout $((T + 120)) "$@" 2>&1 | grep -v amdgpu.ids | tail -4 ) >> $O; }
leg python tools/fuzz_parity.py 100000 $S
FUZZ_BIG_K=1 leg python tools/fuzz_parity.py 100000 $((S + 1))
FUZZ_WIDE_W=1 leg python tools/fuzz_parity.py 100000 $((S + 2))
leg python tools/fuzz_shard.py 100000 $((S + 3))
FUZZ_SHARD_FULL=1 leg python tools/fuzz_shard.py 100000 $((S + 4))
leg python tools/fuzz_devparse.py 100000 $((S + 5))
FUZZ_FASTA=1 leg python tools/fuzz_devparse.py 100000 $((S + 6))
cat $O
