# K1a occupancy experiment: set-table size / LDS padding of k_minimizer_fast (rebuilds hulk_minimizer.o on the box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/k1a
B="python bench.py --no-cpu-baseline --no-cold --single-pass"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value"
for v in "128 2048" "128 0" "64 0" "32 0"; do
  set -- $v
  rm -f hulk_amd/csrc/hulk_minimizer.o
  make -C hulk_amd/csrc CXXFLAGS="$FLAGS -DHULK_FAST_TAB=$1 -DHULK_FAST_PAD=$2" > /dev/null 2>&1
  for rep in 1 2; do
  HULK_NO_OVERLAP=1 $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('tab $1 pad $2 serial', round(d['value']/1e9,4), 'k1a', round(d['roofline']['avg_launch_us'],1), d['sketch_md5'][:8])"
  done
  $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('tab $1 pad $2 overlap', round(d['value']/1e9,4), 'k1a', round(d['roofline']['avg_launch_us'],1), d['sketch_md5'][:8])"
done
rm -f hulk_amd/csrc/hulk_minimizer.o; make -C hulk_amd/csrc > /dev/null 2>&1
