cd $GRAFT_REPO_ROOT
for cfg in "--config montecarlo4096" "--config montecarlo4096 --candidates-per-gpu 128"; do
timeout 1200 python bench.py $cfg --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['config']['workload'][:60], {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_iters_max','plan_status_ok','plans_per_s','plan_setup_ms']}, 'pen_us', round(d['roofline']['avg_kernel_us'],1), 'frac', round(d['roofline']['frac'],4))"
done
