# the whole GPU suite + smoke on the current build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 600 > gpurun_out/suite.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/suite.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/suite.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
