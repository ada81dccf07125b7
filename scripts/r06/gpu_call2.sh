# round 6, second call: the penalty integrator's instruction diet build against build, its parity tests, the service-protocol probe, a bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_gpu_header.py tests/test_golden.py -m gpu -q -p no:cacheprovider --timeout 500 > gpurun_out/r06_tests2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests2.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests2.log | head -30
timeout 600 python scripts/r06/penalty_ab.py > gpurun_out/r06_penalty_ab.jsonl 2> gpurun_out/penalty_ab.err; echo "penalty ab rc=$?"; cut -c1-420 gpurun_out/r06_penalty_ab.jsonl; tail -3 gpurun_out/penalty_ab.err
timeout 120 scripts/micro/service_probe 200 > gpurun_out/r06_service_probe.jsonl 2>&1; echo "service probe rc=$?"; cat gpurun_out/r06_service_probe.jsonl
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_call2.json 2> gpurun_out/bench2.err; echo "bench rc=$?"; tail -2 gpurun_out/bench2.err | cut -c1-300
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_call2.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate']})
print(json.dumps(d.get('boundary_call_us'))[:1800])
print(json.dumps(d['roofline'].get('large_batch'))[:900])
PY
