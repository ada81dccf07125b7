cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_ms_device','plan_us_per_round','plan_ms_one_candidate','plan_status_ok']})"; }
for i in 1 2 3; do run FRX_RESIDENT_CMD_STRIDE=1; run FRX_RESIDENT_CMD_STRIDE=4; done
