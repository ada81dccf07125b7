# round 3, call 3: resident-kernel suite + profile on the build with the deferred result post and the LDS-square pass B; penalty kernel W sweep; multi probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/r03/multi_probe.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/tests_c3.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/tests_c3.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests_c3.log | head -40
cat gpurun_out/direction_pin_s*.json 2>/dev/null | tr -d '\n '; echo
for b in 32 1; do timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/r03_c3_budget_B$b.json 2>&1; head -42 gpurun_out/r03_c3_budget_B$b.json | tr -d '\n '; echo; done
timeout 600 python scripts/r03/penalty_waves.py 2>&1 | tee gpurun_out/r03_penalty_waves.jsonl | tail -40
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>gpurun_out/bench_c3.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','plan_ms','plan_ms_device','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_status_ok','plan_objective_min','plan_ms_per_stage_path']}, d['roofline']['stage_kernels_us'])"; tail -2 gpurun_out/bench_c3.err
