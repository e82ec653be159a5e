cd $GRAFT_REPO_ROOT
for nk in 131072 262144 524288 1048576; do
  echo "HULK_NIB_KEYS=$nk"
  HULK_NIB_KEYS=$nk python tools/run_config.py --k 31 --S 1024 --decay 0.02 --reads 16000000 --interval 100000 --batch 16 | cut -c1-150
done
