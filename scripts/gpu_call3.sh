cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
timeout 120 python -m pytest tests/test_gpu_resident.py -x -q -s > gpurun_out/resident_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/resident_tests.txt
tail -12 gpurun_out/resident_tests.txt
timeout 200 python scripts/resident_profile.py 32 64 16 400 > gpurun_out/rprof_B32.json 2>&1; cat gpurun_out/rprof_B32.json
timeout 200 python scripts/resident_profile.py 1 64 16 400 > gpurun_out/rprof_B1.json 2>&1; cat gpurun_out/rprof_B1.json
