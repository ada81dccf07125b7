cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_resident.py tests/test_gpu_multi.py -m gpu -q -x -p no:cacheprovider --timeout 120 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/ab_tests.log | head -20
timeout 300 python scripts/r03/ab_plan.py FRX_RESIDENT_FORWARD 0 1 3 | tee gpurun_out/r04_ab5_forward.json
timeout 120 python scripts/r04/round_gaps.py 32 3000 240 2>&1 | grep -E "rounds_in|adj_end|confirmed|predicted"
