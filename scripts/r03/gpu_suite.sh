cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 300 > gpurun_out/suite.log 2>&1
grep -E "passed|failed|error" gpurun_out/suite.log | tail -3
