cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --no-cpu-baseline 2>&1 | tail -2
