# round 5, call 1: the GPU suite on the working tree (new tests included), then build against build on this box:
#   ab_r04 = round 4's final tree, ab_advice = + advisor fixes / host budget, ab_pen = + penalty pre-reject and zero terms, . = + adjoint in front of the poll
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 400 --durations=8 > gpurun_out/tests1.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests1.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests1.log | head -30
AB_KAPPA48=1 timeout 900 python scripts/r05/ab_all.py ab_r04 ab_advice ab_pen . ab_fwd 3 > gpurun_out/ab1.jsonl 2> gpurun_out/ab1.err; tail -6 gpurun_out/ab1.jsonl | cut -c1-900
timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/r05a_round_gaps_B32.txt 2>&1; tail -26 gpurun_out/r05a_round_gaps_B32.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a_bench_driver_form.json 2> gpurun_out/bench1.err; echo "bench rc=$?"; tail -2 gpurun_out/bench1.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05a_bench_driver_form.json').read().strip().splitlines()[-1]); r = d['roofline']
    keys = ['value','ms_per_step','ms_per_step_host_wall','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','plan_ms_per_stage_path','plan_mailbox_threads','plan_host_cpus']
    print({k: d.get(k) for k in keys}); print(r['stage_kernels_us'], 'frac', r['frac'], 'large', r['large_batch'])
    print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k not in ('sample','note')})
except Exception as e: print('bench parse failed', e)
PY
