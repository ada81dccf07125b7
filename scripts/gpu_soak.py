"""Soak of the resident round kernel: repeated plans must be bit-identical, with and without optimistic acceptance, across batch shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, frx_import
import fast_racing_amd as frx, fast_racing_amd.scenario as sc
for (B, N, gates, kappa, reps) in [(32, 64, 16, 16, 4), (8, 32, 8, 8, 6), (3, 64, 16, 48, 3), (5, 40, 10, 16, 3), (16, 48, 12, 12, 3)]:
    cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
    p = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = p.initial_guess()
    ref = None
    for r in range(reps):
        res = p.optimize(1e-6, x0=x0)
        key = (res["x"].tobytes(), res["status"].tobytes(), res["objective"].tobytes())
        if ref is None: ref = key
        assert key == ref, f"plan {r} of shape {(B, N, kappa)} differs from plan 0"
    print(B, N, kappa, "resident", res["resident"], "device_status", res["device_status"], "rounds", res["rounds"], "ms %.1f" % res["ms_total"],
          "status ok", int((res["status"] >= 0).sum()), "of", B, "min objective %.6f" % res["objective"].min(), flush=True)
    p.close()
