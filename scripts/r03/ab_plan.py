"""A/B of one environment switch on the production plan, alternating in one process on one box: python ab_plan.py VAR A B [reps] [B]."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
out = {}
for B in ([int(sys.argv[5])] if len(sys.argv) > 5 else [32, 1]):
    cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
    x0 = prob.initial_guess()
    tol = sc.ZHANGJIAJIE["opt_rel_tol"]
    prob.optimize(tol, x0=x0, max_iterations=50)
    res = {va: [], vb: []}
    ref = None
    for i in range(reps):
        for v in (va, vb):
            os.environ[var] = v
            r = prob.optimize(tol, x0=x0)
            res[v].append(round(1e3 * r["ms_total"] / r["rounds"], 3))
            if ref is None: ref = r["x"].copy()
            assert np.array_equal(ref, r["x"]) and r["device_status"] == 0
    out[f"B={B}"] = {f"{var}={v}": {"us_per_round": res[v], "median": float(np.median(res[v]))} for v in (va, vb)}
    prob.close()
print(json.dumps(out))
