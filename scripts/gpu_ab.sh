cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
run() { env "$@" timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','plan_ms','plan_rounds','plan_iters_max','plan_status_ok','plan_objective_min']}, 'us/round %.1f' % (1e3*d['plan_ms']/d['plan_rounds']))"; }
for v in "$@"; do run $v; done
