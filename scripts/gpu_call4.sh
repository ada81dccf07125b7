cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FRX_ROUND_TIMEOUT_MS=3000
for th in 1 4 8 16 32; do
  FRX_RESIDENT_HOST_THREADS=$th timeout 100 python scripts/resident_profile.py 32 64 16 400 2>&1 | python -c "
import sys,json
txt=sys.stdin.read(); j=json.loads(txt[:txt.rindex('}\n{')+1]); print('host threads $th: us/round', round(j['us_per_round_wall'],1), 'wait_host', j['leader']['wait_host'])"
done
timeout 400 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('gpurun_out/bench_r2b.json')); print({k:j[k] for k in j if k.startswith('plan') or k in ('value','ms_per_step')})"
