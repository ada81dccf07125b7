cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider -s > gpurun_out/tests_c9.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/tests_c9.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests_c9.log | head -30; grep "rounds_on_predicted" gpurun_out/tests_c9.log
timeout 400 python scripts/r03/hostwait_probe.py 2>&1 | head -5
for b in 32 1; do timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/r03_c9_budget_B$b.json 2>&1; head -44 gpurun_out/r03_c9_budget_B$b.json | tr -d '\n ' | cut -c1-900; echo; done
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>gpurun_out/bench_c9.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','plan_ms','plan_ms_device','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_status_ok','plan_objective_min','plan_ms_per_stage_path']})"; tail -2 gpurun_out/bench_c9.err
