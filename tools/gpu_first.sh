set -x
cd $GRAFT_REPO_ROOT
rocminfo | grep -E "gfx|Marketing" | head -4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -30
