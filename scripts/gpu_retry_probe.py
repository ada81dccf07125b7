import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, frx_import
import fast_racing_amd as frx, fast_racing_amd.scenario as sc
for (B, N, gates, kappa) in [(16, 48, 12, 12), (16, 48, 12, 16), (8, 48, 12, 12), (32, 64, 16, 16), (32, 32, 8, 8), (12, 56, 14, 16), (20, 24, 6, 8)]:
    cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
    p = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = p.initial_guess()
    res = p.optimize(1e-6, x0=x0)
    bad = [(int(b), int(s), int(res["iters"][b])) for b, s in enumerate(res["status"]) if s < 0]
    print((B, N, kappa), "resident", res["resident"], "rounds", res["rounds"], "ms %.1f" % res["ms_total"], "failed on the resident kernel (candidate, status, iterations):", bad, flush=True)
    p.close()
