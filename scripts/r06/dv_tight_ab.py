#!/usr/bin/env python3
"""History rows of the per-stage L-BFGS as long as the vector needs (n + 2 rounded up to 16 doubles) against the full 64 W E of round 5 (FRX_DV_TIGHT=0), in one process:
k_lbfgs_pre alone (frx_dv_selftest: error against a host two-loop recursion, us per advance with a full history) and a per-stage plan (bit-identical iterates expected)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
for n, B in ((641, 32), (641, 256), (641, 512), (200, 256), (1030, 128)):
    row = {"n": n, "candidates": B}
    for tight in ("0", "1"):
        os.environ["FRX_DV_TIGHT"] = tight
        best = None
        for _ in range(3):
            err, us = frx.dv_selftest(n, B=B, m=128, iters=160)
            best = us if best is None else min(best, us)
        row["tight" if tight == "1" else "full"] = {"max_rel_err": err, "us_per_advance": round(best, 2)}
    row["speedup"] = round(row["full"]["us_per_advance"] / row["tight"]["us_per_advance"], 3)
    print(json.dumps(row), flush=True)
# a per-stage plan both ways
B0, N, gates, kappa = sc.CONFIGS["headline"]
cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(8)]
res = {}
for tight in ("0", "1"):
    os.environ["FRX_DV_TIGHT"] = tight
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    prob.set_resident(False)
    r = prob.optimize(1e-6, x0=prob.initial_guess(), max_iterations=600)
    res[tight] = (hashlib.sha1(r["x"].tobytes()).hexdigest()[:12], int(r["rounds"]), round(r["ms_total"], 2))
    prob.close()
print(json.dumps({"per_stage_plan_8_candidates_600_iterations": {"full": res["0"], "tight": res["1"], "bit_identical": res["0"][0] == res["1"][0]}}))
