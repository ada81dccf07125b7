# A/B of the working tree against a variant built with scripts/r04/make_variant.sh (e.g. `make_variant.sh base HEAD`), on ONE box:
#   gpurun -- 'bash scripts/r04/gpu_ab.sh [variant-name] [tag]'
# resident tests first (direction pin, size classes, cross-XCD form, work queue), then us per round of the headline plan build against build in
# alternating processes (32 candidates, one candidate), then the timeline of one accepted round of the working tree.
cd $GRAFT_REPO_ROOT
VAR=${1:-base}; TAG=${2:-ab}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 300 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/ab_tests.log | head -20
timeout 400 python scripts/r03/ab_libs.py ab_$VAR . 3 32 | tee gpurun_out/r04_${TAG}_B32.json
timeout 300 python scripts/r03/ab_libs.py ab_$VAR . 2 1 | tee gpurun_out/r04_${TAG}_B1.json
timeout 120 python scripts/r04/round_timeline.py 32 3000 3 2>&1 | grep -v member3 | head -70
