import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frx_import  # noqa
import fast_racing_amd as frx, fast_racing_amd.scenario as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cands = [sc.make_candidate(b, 64, 16) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
r = prob.optimize(sc.ZHANGJIAJIE["opt_rel_tol"])
bad = np.where(r["status"] < 0)[0]
print("failed:", [(int(b), int(r["status"][b]), int(r["iters"][b]), int(r["evals"][b]), float(r["objective"][b])) for b in bad])
it = r["iters"]; print("iters percentiles 50/90/99/max", np.percentile(it, [50, 90, 99, 100]))
