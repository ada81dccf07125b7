cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 125 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 100 --ignore=tests/test_gpu_resident.py > gpurun_out/rest.log 2>&1; grep -E "passed|failed|rror" gpurun_out/rest.log | tail -3
