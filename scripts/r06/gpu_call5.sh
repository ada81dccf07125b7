# round 6, fifth call: the two modes of a process and where its mailbox pages live - alternating processes with the allocation scope on (default) and off (round 5's behaviour)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 python scripts/r06/mode_probe.py 1 | sed 's/^{/{"numa_alloc": "device node (default)", /' | cut -c1-260
  timeout 100 python scripts/r06/mode_probe.py 1 FRX_NUMA_ALLOC=0 | sed 's/^{/{"numa_alloc": "wherever the caller runs (FRX_NUMA_ALLOC=0)", /' | cut -c1-260
done > gpurun_out/r06_mode_probe_ab.jsonl 2> gpurun_out/mode_ab.err
cat gpurun_out/r06_mode_probe_ab.jsonl | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_multi.py tests/test_takeover.py -m gpu -q -p no:cacheprovider --timeout 500 > gpurun_out/r06_tests5.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests5.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests5.log | head -20
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_call5.json 2> gpurun_out/bench5.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_call5.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','sclk_mhz_under_latency_bound_fp64_load','plan_kilocycles_per_round']})
print(json.dumps(d.get('boundary_call_us'))[:1200])
PY
