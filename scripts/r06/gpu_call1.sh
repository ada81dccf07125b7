# round 6, first call: the new tests (reference GPU header through the drop-in, full Monte-Carlo share, device-form failure path), the whole GPU suite, a bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_reference_gpu_header.py tests/test_gpu_parity.py::test_device_form_notices_an_expired_wait_by_itself tests/test_gpu_parity.py::test_one_launch_evaluation_fails_loudly_and_the_handle_stays_usable "tests/test_gpu_configs.py::test_config4_full_share_of_one_gpu" tests/test_takeover.py -m gpu -q -p no:cacheprovider --timeout 800 -s -x > gpurun_out/r06_new_tests.log 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_new_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_new_tests.log | head -30
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 --durations=8 > gpurun_out/r06_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests.log | head -30
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_call1.json 2> gpurun_out/bench1.err; echo "bench rc=$?"; tail -2 gpurun_out/bench1.err | cut -c1-300
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_call1.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate']})
print(json.dumps(d.get('boundary_call_us'))[:1500])
PY
