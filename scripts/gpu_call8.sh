cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.txt
tail -4 gpurun_out/gpu_tests.txt
