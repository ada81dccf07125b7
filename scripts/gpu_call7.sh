cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -3
for w in 3 4; do FRX_PENALTY_WAVES=$w timeout 200 python scripts/kernel_sweep.py --batches 32,1024,4096 --states it60 --reps 30 2>&1 | sed "s/^/waves=$w /"; done
timeout 900 bash scripts/gpu_pmc_round2.sh > gpurun_out/pmc2.log 2>&1; tail -3 gpurun_out/pmc2.log
timeout 500 python bench.py > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_r2d.err
python -c "
import json; j=json.load(open('gpurun_out/bench_r2d.json')); print({k:j[k] for k in j if k.startswith('plan') or k in ('value','ms_per_step')}); r=j['roofline']; print(r['kernel'], r['frac'], r['stage_kernels_us'], r['penalty']['fp64'], r['penalty']['large_batch']); print(r['states'])"
