"""Where do the gradients of the one-launch evaluation and of the stage kernels differ?  (diagnostic)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B, N, gates, kappa = sc.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "headline"]
cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
x0 = prob.initial_guess()
xs = prob.optimize(sc.ZHANGJIAJIE["opt_rel_tol"], x0=x0, max_iterations=60)["x"]
lay = prob.layout() if hasattr(prob, "layout") else None
for nm, x in (("x0", x0), ("xs", xs)):
    prob.set_eval_fused(True); f1, g1 = prob.objective(x)
    prob.set_eval_fused(False); f3, g3 = prob.objective(x)
    d = np.abs(g1 - g3); idx = np.argsort(-d)[:12]
    print(nm, "max|g|", float(np.max(np.abs(g3))), "n differing", int(np.sum(d > 0)), "of", d.size, "max rel", float(np.max(d / (np.abs(g3) + 1e-300))))
    print("  worst:", [(int(i), float(g1[i]), float(g3[i])) for i in idx[:6]])
    xo = prob.x_off if hasattr(prob, "x_off") else None
    if xo is not None:
        xo = np.asarray(xo)
        for i in idx[:6]:
            c = int(np.searchsorted(xo, i, side="right") - 1)
            print("   index", int(i), "candidate", c, "offset in candidate", int(i - xo[c]), "of", int(xo[c + 1] - xo[c]))
