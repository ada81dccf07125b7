cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 300 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/ab_tests.log | head -20
timeout 400 python scripts/r03/ab_libs.py ab_base . 3 32 | tee gpurun_out/r04_ab20_B32.json
timeout 300 python scripts/r03/ab_libs.py ab_base . 2 1 | tee gpurun_out/r04_ab20_B1.json
timeout 120 python scripts/r04/round_timeline.py 32 3000 3 2>&1 | grep -v member3 | head -70
