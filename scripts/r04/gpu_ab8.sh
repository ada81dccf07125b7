cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_resident.py tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout 120 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/ab_tests.log | head -20
timeout 300 python scripts/r03/ab_plan.py FRX_RESIDENT_SPECULATE 2 2 2 | tee gpurun_out/r04_ab8.json
timeout 400 python scripts/r03/ab_libs.py ab_pre ab_numa . 2 32 | tee gpurun_out/r04_ab8_B32.json
timeout 300 python scripts/r03/ab_libs.py ab_pre ab_numa . 2 1 | tee gpurun_out/r04_ab8_B1.json
