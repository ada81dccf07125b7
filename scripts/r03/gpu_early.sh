cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 100 > gpurun_out/res.log 2>&1; grep -E "passed|failed|rror" gpurun_out/res.log | tail -3
timeout 200 python scripts/r03/ab_plan.py FRX_RESIDENT_EARLY_PASS 0 1 4
