import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
def run(B, G=None, spec=0):
    for k, v in (("FRX_RESIDENT_G", G), ("FRX_RESIDENT_SPECULATE", spec)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
    x0 = prob.initial_guess()
    prob.optimize(1e-6, x0=x0, max_iterations=20)
    os.environ["FRX_RESIDENT_PROF"] = "1"
    r = prob.optimize(1e-6, x0=x0, max_iterations=800)
    del os.environ["FRX_RESIDENT_PROF"]
    pr = prob.resident_profile()
    rounds = int(r["evals"][0]); h = prob.last_host_wait_hist.sum(axis=0)
    print(json.dumps({"B": B, "G": r["resident"], "speculate": spec, "us_per_round": round(1e3 * r["ms_total"] / r["rounds"], 2), "wait_host": round(float(pr[0, 0, 0]) / rounds, 2), "hist": [int(v) for v in h[:9]]}), flush=True)
    prob.close()
for B, G in ((32, None), (16, None), (16, 16), (8, None), (8, 16), (8, 32), (4, None), (4, 32), (1, None), (1, 32), (1, 64), (1, 128), (1, 256)):
    try: run(B, G)
    except Exception as e: print("B", B, "G", G, "failed", str(e)[:100])
