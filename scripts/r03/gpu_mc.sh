cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 python bench.py --config montecarlo4096 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline > gpurun_out/r03_bench_montecarlo4096.json 2> gpurun_out/bench_mc.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_montecarlo4096.json').read().strip().splitlines()[-1])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('plan') or k in ('value','ms_per_step','work_queue_equals_default_path_status')})
PY
