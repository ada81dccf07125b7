# round 2, call 2: first run of the resident round kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
timeout 600 python -m pytest tests/test_gpu_resident.py -x -q -s > gpurun_out/resident_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/resident_tests.txt
tail -40 gpurun_out/resident_tests.txt
