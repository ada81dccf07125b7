# round 2, session 2, call 1: parity suite + timings after the penalty-integrator rewrite (LDS views instead of volatile flat loads)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/tests.log
timeout 300 python scripts/kernel_sweep.py --batches 32,1024 --states it60 --reps 200 --full > gpurun_out/sweep_occ3.log 2>&1; cat gpurun_out/sweep_occ3.log | tail -3
FRX_PENALTY_WAVES=4 timeout 300 python scripts/kernel_sweep.py --batches 32,1024 --states it60 --reps 200 > gpurun_out/sweep_occ4.log 2>&1; tail -3 gpurun_out/sweep_occ4.log
timeout 300 python scripts/resident_profile.py 1 64 16 3000 > gpurun_out/budget_B1.json 2>&1; head -48 gpurun_out/budget_B1.json
timeout 300 python scripts/resident_profile.py 32 64 16 400 > gpurun_out/budget_B32.json 2>&1; head -12 gpurun_out/budget_B32.json
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print({k:d[k] for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate']}, d['roofline']['stage_kernels_us'], d['roofline']['penalty']['large_batch'])"
