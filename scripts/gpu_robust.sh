# odd shapes, the 65..128-piece resident case, the Monte-Carlo share and the multi-process bench on one device
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/gpu_stress.py 2>&1 | tail -4
timeout 300 python - <<'PY' 2>&1 | tail -6
import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, frx_import
import fast_racing_amd as frx, fast_racing_amd.scenario as sc
# 100 pieces: two knots per lane; resident when the variable count allows (n >= 2 m) - compare with the per-stage rounds
c = [sc.make_candidate(5, 100, 25, perturb_id=i) for i in range(2)]
p = frx.Problem(c, sc.ZHANGJIAJIE, qd_intervals=8)
x0 = p.initial_guess()
ra = p.optimize(1e-6, x0=x0, max_iterations=300)
p.set_resident(False)
rb = p.optimize(1e-6, x0=x0, max_iterations=300)
print("N=100: resident", ra["resident"], "device_status", ra["device_status"], "ms %.1f vs per-stage %.1f" % (ra["ms_total"], rb["ms_total"]), "objective rel diff", np.abs(ra["objective"] / rb["objective"] - 1).max())
p.close()
PY
timeout 900 python bench.py --config montecarlo4096 --steps 20 --warmup 5 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('montecarlo share', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_status_ok','plans_per_s']}, d['roofline']['stage_kernels_us'])"
FRX_BENCH_DEVICE=0 FRX_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --large-batch 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks on one device', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','n_gpus','plan_ms','winner_id','winner_rank','plan_status_ok']})"
