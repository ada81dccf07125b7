#!/usr/bin/env python3
"""The three stage kernels over batch sizes (at Monte-Carlo scale the knot kernels are 59 % of an evaluation).  The headline batch replicated
(every replica owns its data in HBM); HIP events inside the library around back-to-back launches of one kernel at a time (frx_eval_stage_times).
  python scripts/r06/knot_sweep.py [batches, comma separated] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc

batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "32,64,128,256,384,512,768,1024,2048").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B0, N, gates, kappa = sc.CONFIGS["headline"]
base = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B0)]
p0 = frx.Problem(base, sc.ZHANGJIAJIE, qd_intervals=kappa)
xs = p0.optimize(1e-6, x0=p0.initial_guess(), max_iterations=60)["x"]
for B in batches:
    rep, rem = divmod(B, B0)
    cands = base * rep + base[:rem]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    xb = np.concatenate([xs] * rep + [xs[:prob.x_off[rem]]]) if rem else np.concatenate([xs] * rep)
    best = None
    for _ in range(3):
        t = prob.stage_times(xb, reps)
        best = t if best is None else {k: min(best[k], t[k]) for k in t}
    # the whole evaluation as the library launches it: three stage launches against the solo form (one workgroup per candidate, one launch; frx_solo_kernel.hpp)
    ev = {}
    fused = prob.eval_fused()
    if fused: prob.set_eval_fused(False)
    for name, mode in (("three_launches", 0), ("solo", 2)):
        if mode and not prob.solo_applies(): continue
        prob.set_eval_solo(mode)
        ev[name] = min(prob.eval_launch_time(xb, reps) for _ in range(3))
    f3 = g3 = None
    if "solo" in ev:
        prob.set_eval_solo(0); f3, g3 = prob.objective(xb)
        prob.set_eval_solo(2); f1, g1 = prob.objective(xb)
        ev["solo_bit_identical"] = bool(np.array_equal(f1, f3) and np.array_equal(g1, g3))
        ev["solo_workgroups_per_cu"] = prob.eval_solo()
    print(json.dumps({"candidates": B, **{k + "_us": round(v, 2) for k, v in best.items()}, "sum_us": round(sum(best.values()), 2),
                      "eval_us": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ev.items()},
                      "ns_per_candidate": {k: round(1e3 * v / B, 1) for k, v in best.items()}}), flush=True)
    prob.close()
