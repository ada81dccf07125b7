import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frx_import import frx
from fast_racing_amd import scenario as sc
for kappa, Bs in ((48, (1, 2, 8, 9, 16, 17)), (16, (1, 9, 16, 32)), (24, (1, 4))):
    for B in Bs:
        prob = frx.Problem([sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)], sc.ZHANGJIAJIE, qd_intervals=kappa)
        r = prob.optimize(1e-6, x0=prob.initial_guess(), max_iterations=30)
        print(json.dumps({"kappa": kappa, "B": B, "workgroups_per_candidate": r["resident"], "clusters": r["clusters"], "us_per_round": round(1e3 * r["ms_total"] / r["rounds"], 2)}))
        prob.close()
