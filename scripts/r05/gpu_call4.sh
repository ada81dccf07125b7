# round 5, call 4: the whole GPU suite on the current tree + the bench line in the driver's form (graph-timed steps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 400 --durations=6 > gpurun_out/tests4.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests4.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests4.log | head -30
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r05d_bench_driver_form.json 2> gpurun_out/bench4.err; echo "bench rc=$?"; tail -2 gpurun_out/bench4.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05d_bench_driver_form.json').read().strip().splitlines()[-1]); r = d['roofline']
keys = ['value','ms_per_step','ms_per_step_host_wall','ms_per_step_direct_launches','launch','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','plan_ms_per_stage_path']
print({k: d.get(k) for k in keys}); print(r['stage_kernels_us'], 'frac', r['frac'], 'large', r['large_batch'])
PY
