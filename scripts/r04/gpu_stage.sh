cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider --timeout 120 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/ab_tests.log | head -20
python - <<'PY'
import sys, json, subprocess
child = r'''
import sys; sys.path.insert(0, sys.argv[1])
import numpy as np, torch, time
from frx_import import frx
from fast_racing_amd import scenario as sc
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(32)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess(); x = prob.optimize(1e-6, x0=x0, max_iterations=60)["x"]
st = prob.stage_times(x, reps=300)
xd = torch.from_numpy(x).cuda(); fd = torch.zeros(32, dtype=torch.float64, device="cuda"); gd = torch.zeros(prob.NX, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(50): prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(400): prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), s)
torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 400 * 1e6
print({"stage_us": {k: round(v, 2) for k, v in st.items()}, "eval_us": round(us, 2)})
'''
for rep in range(2):
    for root in ("ab_numa", "."):
        p = subprocess.run([sys.executable, "-c", child, root], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120)
        print(root, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "FAILED")
PY
