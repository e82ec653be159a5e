#!/bin/bash
# ThreadSanitizer run of the host ingest path (worker teams, the three gzip readers, block reader, parser): hulk_ingest.hip is
# compiled as host C++ by ROCm's clang with -fsanitize=thread, linked with tools/tsan/ingest_stubs.cpp and the counting driver
# tools/ubench/parse_rate.c, and run on the files given (default: a synthetic FASTQ as plain / one-member .gz / two-member .gz
# / bgzip-like members are the caller's to supply).  CPU only.  usage: tools/tsan_ingest.sh [file ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/.scratch/tsan; mkdir -p $OUT
CXX=/opt/rocm/lib/llvm/bin/clang++; CC=/opt/rocm/lib/llvm/bin/clang
F="-O1 -g -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
$CXX -x c++ -std=c++17 $F -fPIC -c $ROOT/hulk_amd/csrc/hulk_ingest.hip -o $OUT/ingest_tsan.o
$CXX -x c++ -std=c++17 $F -I$ROOT -c $ROOT/tools/tsan/ingest_stubs.cpp -o $OUT/stubs.o
$CC -O1 -g -fsanitize=thread -I$ROOT/include -c $ROOT/tools/ubench/parse_rate.c -o $OUT/main.o
$CXX -fsanitize=thread -o $OUT/parse_tsan $OUT/main.o $OUT/ingest_tsan.o $OUT/stubs.o -lz -lpthread
if [ $# -eq 0 ]; then
  python3 - <<PY
import gzip, random
random.seed(1)
recs = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(random.choices(b"ACGT", k=150)), b"F" * 150) for i in range(120000))
open("$OUT/t.fq", "wb").write(recs)
open("$OUT/t.fq.gz", "wb").write(gzip.compress(recs, 6))
open("$OUT/t2.fq.gz", "wb").write(gzip.compress(recs[:recs.index(b"\n@", len(recs) // 2) + 1], 1) + gzip.compress(recs, 6))
open("$OUT/tcut.fq.gz", "wb").write(gzip.compress(recs, 6)[:3000000])
PY
  set -- $OUT/t.fq $OUT/t.fq.gz $OUT/t2.fq.gz $OUT/tcut.fq.gz
fi
for f in "$@"; do echo "== $f"; HULK_GZ_PAR_CHUNK=${HULK_GZ_PAR_CHUNK:-262144} $OUT/parse_tsan "$f" 1 2>&1 | tail -12 | cut -c1-240 || true; done
# the FASTA mode: the block cut into pieces parsed side by side (parse_rate's third argument: 1 = --fasta)
python3 - <<PY
import random
random.seed(2)
open("$OUT/t.fa", "wb").write(b"".join(b">c%d\n" % i + b"".join(bytes(random.choices(b"ACGT", k=60)) + b"\n" for _ in range(3000)) for i in range(40)))
PY
echo "== $OUT/t.fa (fasta)"; HULK_INGEST_BLOCK=262144 $OUT/parse_tsan $OUT/t.fa 1 1 2>&1 | tail -12 | cut -c1-240 || true
