import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, frx_import
import fast_racing_amd as frx, fast_racing_amd.scenario as sc
from oracle import binding as ob
# many small candidates, a long one, and a one-interval quadrature
for (B, N, gates, kappa, obst) in [(600, 8, 2, 4, True), (2, 120, 30, 8, False), (3, 16, 4, 1, False)]:
    cands = [sc.make_candidate(3, N, gates, perturb_id=i, obstacles=obst) for i in range(B)]
    p = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = p.initial_guess()
    f, g = p.objective(x0)
    o = ob.Oracle(cands[B - 1], sc.ZHANGJIAJIE, qd_intervals=kappa); o.set_abscissa_mode(False)
    fr, gr = o.objective(x0[p.x_off[B - 1]:p.x_off[B]])
    r = p.optimize(1e-4, max_iterations=300)
    print(B, N, kappa, "f rel err %.1e grad %.1e" % (abs(f[B - 1] - fr) / abs(fr), np.abs(g[p.x_off[B - 1]:p.x_off[B]] - gr).max() / max(np.abs(gr).max(), abs(fr))),
          "plan rounds", r["rounds"], "status ok", int((r["status"] >= 0).sum()), "of", B, "ms %.1f" % r["ms_total"])
    p.close()
