cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_rounds_one_candidate','ms_per_step','value']})"
timeout 100 python scripts/r03/ab_plan.py FRX_RESIDENT_EARLY_PASS 0 1 2 32
