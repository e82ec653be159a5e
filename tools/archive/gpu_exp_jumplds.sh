# C3-shaped rate against the occupancy cap of k_jump_bin (dummy LDS per workgroup: 160 KB / it workgroups per CU = waves per SIMD):
# fewer jump waves leave wave slots to the flush kernels of the batch before (second stream)
cd $GRAFT_REPO_ROOT
for l in 0 22000 26000 32000 40000; do
  echo "HULK_JUMP_LDS=$l"
  for i in 1 2; do HULK_JUMP_LDS=$l python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c60-150; done
done
echo "C2:"; for l in 0 26000; do HULK_JUMP_LDS=$l python bench.py --no-cpu-baseline --no-cold --no-e2e --single-pass 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print($l, d['value'], d['ms_per_step'])"; done
