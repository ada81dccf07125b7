cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/r04_round_gaps_B32.txt 2>&1
timeout 120 python scripts/r04/round_gaps.py 1 3000 240 > gpurun_out/r04_round_gaps_B1.txt 2>&1
tail -4 gpurun_out/r04_round_gaps_B32.txt; tail -3 gpurun_out/r04_round_gaps_B1.txt
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','plan_ms','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate')})"
