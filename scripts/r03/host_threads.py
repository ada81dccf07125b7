"""Resident plan time of the headline batch against the number of mailbox service threads on the host (FRX_RESIDENT_HOST_THREADS)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B, N, gates, kappa = sc.CONFIGS["headline"]
cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
x0 = prob.initial_guess()
prob.optimize(1e-6, x0=x0, max_iterations=50)
for rep in range(2):
    for n in (1, 2, 3, 4, 6, 8, 16, 32):
        os.environ["FRX_RESIDENT_HOST_THREADS"] = str(n)
        r = prob.optimize(1e-6, x0=x0)
        print(json.dumps({"host_threads": n, "plan_ms": round(r["ms_total"], 2), "rounds": r["rounds"], "us_per_round": round(1e3 * r["ms_total"] / r["rounds"], 2), "host_ms": round(r["ms_host"], 2)}), flush=True)
os.environ.pop("FRX_RESIDENT_HOST_THREADS", None)
