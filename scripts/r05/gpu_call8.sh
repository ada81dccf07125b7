# round 5, call 8: slot / pair count of the next step stored at gather time (working tree) against the tree before (ab_head): resident + take-over tests, timelines, build against build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_takeover.py tests/test_gpu_multi.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -3
for v in ab_head . ab_head .; do FRX_ROOT=$v timeout 120 python scripts/r04/round_gaps.py 32 3000 240 2>&1 | grep -E "rounds_in_the_sample|predicted -> member sees ADV|confirmed -> predicted" | cut -c1-150; done
timeout 1200 python scripts/r05/ab_all.py ab_head . 6 > gpurun_out/ab8.jsonl 2> gpurun_out/ab8.err; tail -2 gpurun_out/ab8.jsonl | cut -c1-600
