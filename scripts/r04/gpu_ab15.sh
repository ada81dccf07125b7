cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider --timeout 300 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/ab_tests.log | head -20
timeout 400 python scripts/r03/ab_libs.py ab_base . 3 32 | tee gpurun_out/r04_ab15_B32.json
timeout 300 python scripts/r03/ab_libs.py ab_base . 2 1 | tee gpurun_out/r04_ab15_B1.json
timeout 120 python scripts/r04/round_timeline.py 32 3000 3 2>&1 | grep -v member3 | head -44
python - <<'PY'
import sys; sys.path.insert(0,'.')
from frx_import import frx
from fast_racing_amd import scenario as sc
import numpy as np
cands=[sc.make_candidate(0,64,16,perturb_id=b) for b in range(32)]
p=frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16); x0=p.initial_guess()
print("stage kernels us", p.eval_stage_times(x0, 300))
PY
