# round 5, call 3: the take-over (stragglers of a per-stage batch continue on the resident kernel): tests, then the throughput configuration with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_takeover.py "tests/test_gpu_parity.py::test_lockstep_parity_along_the_whole_optimisation" -m gpu -q -p no:cacheprovider --timeout 600 -s > gpurun_out/tests3.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests3.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests3.log | head -30
grep -E "^\{\"(B\"|candidates|hand_over)" gpurun_out/tests3.log | cut -c1-600
timeout 500 python bench.py --config montecarlo4096 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline > gpurun_out/r05c_bench_montecarlo4096.json 2> gpurun_out/bench_mc3.err; echo "mc rc=$?"; tail -2 gpurun_out/bench_mc3.err | cut -c1-300
python -c "
import json; d=json.loads(open('gpurun_out/r05c_bench_montecarlo4096.json').read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('plan') and not isinstance(v,(list,))})"
