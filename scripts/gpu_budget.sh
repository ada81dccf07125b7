cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/tests.log | head -20
timeout 300 python scripts/resident_profile.py 1 64 16 3000 > gpurun_out/budget_B1.json 2>&1; head -80 gpurun_out/budget_B1.json | tr -d '\n ' ; echo
timeout 300 python scripts/resident_profile.py 32 64 16 3000 > gpurun_out/budget_B32.json 2>&1; head -80 gpurun_out/budget_B32.json | tr -d '\n '; echo
run() { env "$@" timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','plan_ms','plan_ms_device','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_status_ok','plan_objective_min']}, d['roofline']['stage_kernels_us'])"; }
run FRX_X=1
