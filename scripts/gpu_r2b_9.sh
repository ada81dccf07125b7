cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch 0 --candidates-per-gpu $CB 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$CB $*', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['plan_ms','plan_ms_device','plan_rounds','plan_us_per_round','plan_status_ok']})"; }
for CB in 2 4 8 16 32; do run FRX_X=0; run FRX_RESIDENT_SPECULATE=0; done
