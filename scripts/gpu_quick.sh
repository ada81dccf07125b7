cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
run() { env "$@" timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','plan_ms','plan_ms_device','plan_rounds','plan_iters_max','plan_status_ok','plan_objective_min']})"; }
run FRX_LBFGS=device
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dv -o dv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_dv.err
head -7 $R/gpurun_out/prof_dv/dv_kernel_stats.csv | cut -c1-150; rm -f $R/gpurun_out/prof_dv/dv_kernel_trace.csv
