cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_resident.py -m gpu -q -p no:cacheprovider --timeout 120 2>&1 | tail -2
for lv in 1 2; do FRX_RESIDENT_SPECULATE=$lv timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('level $lv', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_rounds_one_candidate']})"; done
for b in 32 1; do timeout 300 python scripts/resident_profile.py $b 64 16 3000 2>&1 | head -14 | tr -d '\n ' | cut -c1-420; echo; done
