cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 60 > gpurun_out/res.log 2>&1; grep -E "passed|failed|rror" gpurun_out/res.log | tail -3
timeout 150 python scripts/r03/ab_libs.py ab_prev . 2 32
