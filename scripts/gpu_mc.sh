cd $GRAFT_REPO_ROOT
run() { env $1 timeout 1200 python bench.py $2 --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_iters_max','plan_status_ok','plans_per_s','plan_objective_min']})"; }
run FRX_SKIP_INACTIVE=1 "--config montecarlo4096"
run FRX_SKIP_INACTIVE=0 "--config montecarlo4096"
run FRX_SKIP_INACTIVE=1 "--config headline"
run FRX_SKIP_INACTIVE=0 "--config headline"
