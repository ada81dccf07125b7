#!/usr/bin/env python3
"""The solo launch at one batch size, three launches beside it: us per evaluation, bit-identity.  (Form 1 of the kernel had measurement switches, FRX_SOLO_DEBUG - e.g. the forward
map without its multiplier stores, 36.7 -> 35.0 us at 512 candidates; form 2 keeps the multipliers in LDS and has none: the second argument is ignored.)
  python scripts/r06/solo_probe.py [candidates]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
switches = ["0"]
B0, N, gates, kappa = sc.CONFIGS["headline"]
base = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B0)]
p0 = frx.Problem(base, sc.ZHANGJIAJIE, qd_intervals=kappa)
xs = p0.optimize(1e-6, x0=p0.initial_guess(), max_iterations=60)["x"]
rep = B // B0
prob = frx.Problem(base * rep, sc.ZHANGJIAJIE, qd_intervals=kappa)
xb = np.concatenate([xs] * rep)
prob.set_eval_solo(0)
f3, g3 = prob.objective(xb)
out = {"candidates": B, "three_launches_us": round(min(prob.eval_launch_time(xb, 100) for _ in range(3)), 2)}
prob.set_eval_solo(2)
for sw in switches:
    os.environ["FRX_SOLO_DEBUG"] = sw
    t = min(prob.eval_launch_time(xb, 100) for _ in range(3))
    f1, g1 = prob.objective(xb)
    out["solo_debug_%s" % sw] = {"us": round(t, 2), "same_bits": bool(np.array_equal(f1, f3) and np.array_equal(g1, g3))}
print(json.dumps(out))
