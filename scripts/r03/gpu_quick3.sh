cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider --timeout 200 2>&1 | tail -3
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_rounds_one_candidate','ms_per_step']}, d['roofline']['stage_kernels_us'])"
for b in 32 1; do timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/round_budget_B$b.json 2>&1; python - <<PY
import json
t=open('gpurun_out/round_budget_B$b.json').read()
d=json.loads(t[:t.index('\n}\n')+2])
print($b, d['us_per_round_wall'], d['leader'], d['forward_stamps'], d['adjoint_stamps'])
PY
done
