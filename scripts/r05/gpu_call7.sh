# round 5, call 7: the two-phase large-batch penalty integrator: parity tests, then one-phase against two-phase at 256 / 1024 / 4096 candidates and kappa = 16 / 48-> one sample per lane only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "penalty or large_batch or two_phase" 2>&1 | tail -3
python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B0, N, gates, kappa = sc.CONFIGS["headline"]
base = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B0)]
small = frx.Problem(base, sc.ZHANGJIAJIE, qd_intervals=kappa)
xs = small.optimize(1e-6, max_iterations=60)["x"]
small.close()
out = {}
for rep in (8, 32, 128):
    big = frx.Problem(base * rep, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x = np.tile(xs, rep)
    r = {}
    for form in ("1", "0", "1", "0"):
        os.environ["FRX_PENALTY_TWOPHASE"] = form
        r.setdefault("two_phase" if form == "1" else "one_phase", []).append(round(min(big.stage_times(x, reps=30)["penalty"] for _ in range(2)), 2))
    os.environ.pop("FRX_PENALTY_TWOPHASE")
    alg = big.algorithmic_bytes()
    r["hbm_frac_two_phase"] = round(alg / (min(r["two_phase"]) * 1e-6) / 8e12, 4); r["hbm_frac_one_phase"] = round(alg / (min(r["one_phase"]) * 1e-6) / 8e12, 4)
    out[f"{B0 * rep} candidates"] = r
    big.close()
print(json.dumps(out))
open("gpurun_out/r05_penalty_two_phase.json", "w").write(json.dumps(out) + "\n")
PY
