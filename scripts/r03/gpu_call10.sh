cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 240 --durations=6 > gpurun_out/tests_c10.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/tests_c10.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests_c10.log | head -40; grep -A7 "slowest" gpurun_out/tests_c10.log | tail -7
