cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_ms_device','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_status_ok','plan_objective_min']})"; }
for v in FRX_RESIDENT_POLL=0 FRX_RESIDENT_POLL=1 FRX_RESIDENT_POLL=2 FRX_RESIDENT_POLL=3; do run $v; done
for v in FRX_RESIDENT_HOST_THREADS=1 FRX_RESIDENT_HOST_THREADS=4 FRX_RESIDENT_HOST_THREADS=8 FRX_RESIDENT_HOST_THREADS=16; do run $v; done
run FRX_RESIDENT_SPECULATE=0
