"""A/B of one environment switch on the per-stage plan of a Monte-Carlo share (default 512 scenarios): python ab_per_stage.py VAR A B [reps] [B]."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
B = int(sys.argv[5]) if len(sys.argv) > 5 else 512
cands = [sc.make_candidate(b, 64, 16) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
prob.set_resident(0)
x0 = prob.initial_guess()
tol = sc.ZHANGJIAJIE["opt_rel_tol"]
prob.optimize(tol, x0=x0, max_iterations=30)
res = {va: [], vb: []}
ref = None
for i in range(reps):
    for v in (va, vb):
        os.environ[var] = v
        r = prob.optimize(tol, x0=x0, max_iterations=60000)
        res[v].append({"plan_ms": round(r["ms_total"], 1), "device_ms": round(r["ms_device"], 1), "host_ms": round(r["ms_host"], 1), "rounds": r["rounds"], "plans_per_s": round(1e3 * B / r["ms_total"], 1)})
        if ref is None: ref = (r["x"].copy(), r["status"].copy())
        assert np.array_equal(ref[0], r["x"]) and np.array_equal(ref[1], r["status"])
print(json.dumps({"candidates": B, **{f"{var}={v}": res[v] for v in (va, vb)}}))
prob.close()
