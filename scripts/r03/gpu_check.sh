cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 300 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_rounds_one_candidate','ms_per_step']}, d['roofline']['stage_kernels_us'])"; done
