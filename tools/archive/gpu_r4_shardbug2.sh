#!/bin/bash
O=gpurun_out; mkdir -p $O
HULK_SHARD_DEBUG=1 timeout 300 python tools/fuzz_shard.py 30 16 > $O/sb2.out 2> $O/sb2.err; echo "rc=$?"
tail -2 $O/sb2.out | cut -c1-300
grep -n "hulk shard\|callback" $O/sb2.err | grep -B30 -A10 "stale" | head -80 | cut -c1-200
