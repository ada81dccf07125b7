# round 4: kernel steps measured build against build on one box; parity tests of the working tree first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 120 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/ab_tests.log | head -20
timeout 500 python scripts/r03/ab_libs.py ab_base ab_ab . 3 32 | tee gpurun_out/r04_ab1_B32.json
timeout 400 python scripts/r03/ab_libs.py ab_base ab_ab . 2 1 | tee gpurun_out/r04_ab1_B1.json
python - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(32)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess(); x = prob.optimize(1e-6, x0=x0, max_iterations=60)["x"]
print("stage kernels us", prob.stage_times(x, reps=300))
PY
