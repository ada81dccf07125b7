"""BASELINE configs[0] (ONE candidate x 64 pieces x the stock kappa = 48): 64 wave-tasks of the penalty integral on (G - 1) x 4 waves - does a cluster of 17 or 18
workgroups (one pass) beat the default 16 (two passes)?   FRX_RESIDENT_G=n python scripts/r05/plumbing_g_probe.py"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
prob = frx.Problem([sc.make_candidate(0, 64, 16, perturb_id=0)], sc.ZHANGJIAJIE, qd_intervals=48)
x0 = prob.initial_guess()
tol = sc.ZHANGJIAJIE["opt_rel_tol"]
prob.optimize(tol, x0=x0, max_iterations=50)
v = []
for i in range(3):
    r = prob.optimize(tol, x0=x0)
    v.append(round(1e3 * r["ms_total"] / r["rounds"], 3))
print(json.dumps({"FRX_RESIDENT_G": os.environ.get("FRX_RESIDENT_G"), "us_per_round": v, "rounds": int(r["rounds"]), "plan_ms": round(r["ms_total"], 2), "resident": r["resident"], "clusters": r["clusters"],
                  "objective": float(r["objective"][0]), "status": int(r["status"][0]), "x_sha": hashlib.sha1(np.ascontiguousarray(r["x"]).tobytes()).hexdigest()[:12]}))
