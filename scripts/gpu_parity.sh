cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stagewise or fixed_total or accumulates or short_run" 2>&1 | tail -30
