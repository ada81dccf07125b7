set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E 'gfx|Compute Unit' | head -4
nproc
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s 2>&1 | tail -40
