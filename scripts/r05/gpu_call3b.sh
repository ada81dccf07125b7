cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_takeover.py -m gpu -q -p no:cacheprovider --timeout 600 -s > gpurun_out/tests3b.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests3b.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests3b.log | head -30
grep -E "^\{\"(B\"|candidates|hand_over)" gpurun_out/tests3b.log | cut -c1-900
