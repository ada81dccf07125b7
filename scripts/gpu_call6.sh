cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.txt
tail -5 gpurun_out/gpu_tests.txt
timeout 400 python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('gpurun_out/bench_r2c.json')); print({k:j[k] for k in j if k.startswith('plan') or k in ('value','ms_per_step')})"
