cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.txt
tail -5 gpurun_out/gpu_tests.txt; grep -h "^route" gpurun_out/gpu_tests.txt | head
