#!/bin/bash
# Does a pass keep getting faster beyond 20 steps?  (clock ramp under continuous load)
mkdir -p gpurun_out/firstpass; O=gpurun_out/firstpass/long.txt; : > $O
echo "== 200 synchronised steps" >> $O
HULK_BENCH_STEPTIMES=2 HULK_BENCH_PREWARM_S=0 python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-cold --no-e2e --single-pass 2>&1 | grep -v amdgpu.ids | cut -c1-2600 >> $O
for K in 20 100 400; do
  echo "== --steps $K --warmup 3 (default pre-warm)" >> $O
  python bench.py --steps $K --warmup 3 --no-cpu-baseline --no-cold --no-e2e --single-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['k_jump_bin']['avg_launch_us'])" >> $O
done
( while true; do cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | tr '\n' ' '; echo; sleep 0.05; done > gpurun_out/firstpass/sclk.txt ) &
SP=$!
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-cold --no-e2e --single-pass > /dev/null 2>&1
kill $SP
sort gpurun_out/firstpass/sclk.txt | uniq -c | sort -rn | head -12 >> $O
cat $O
