cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py tests/test_golden.py -x -q > gpurun_out/tests5.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/tests5.txt
tail -4 gpurun_out/tests5.txt
timeout 200 python scripts/resident_profile.py 32 64 16 400 > gpurun_out/rprof_B32.json 2>&1
timeout 200 python scripts/resident_profile.py 1 64 16 400 > gpurun_out/rprof_B1.json 2>&1
