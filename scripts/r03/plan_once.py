"""One full 32-candidate headline plan (and one single-candidate plan) on the production kernel: ms and us per round."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frx_import import frx
from fast_racing_amd import scenario as sc
out = {}
for B in (32, 1):
    cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
    x0 = prob.initial_guess()
    tol = sc.ZHANGJIAJIE["opt_rel_tol"]
    prob.optimize(tol, x0=x0, max_iterations=50)
    rs = [prob.optimize(tol, x0=x0) for _ in range(2)]
    out[f"B={B}"] = {"plan_ms": [round(r["ms_total"], 2) for r in rs], "rounds": int(rs[0]["rounds"]), "us_per_round": [round(1e3 * r["ms_total"] / r["rounds"], 2) for r in rs]}
    prob.close()
print(json.dumps(out))
