cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 120 > gpurun_out/res.log 2>&1; grep -E "passed|failed|rror" gpurun_out/res.log | tail -3
for b in 1 32; do FRX_PROFILE_MODE=2 timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/p2_B$b.json 2>&1; python - <<PY
import json
t=open('gpurun_out/p2_B$b.json').read()
d=json.loads(t[:t.index('\n}\n')+2])
print($b, d['us_per_round_wall'], d['leader'])
PY
done
for i in 1 2; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_rounds_one_candidate']})"; done
