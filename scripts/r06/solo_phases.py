#!/usr/bin/env python3
"""Cycle stamps of candidate 0 inside the solo launch (frx_solo_kernel.hpp): entry, forward map done, penalty phase done, adjoint done - at several batch sizes
(the phases of one workgroup stretch with its neighbours on the CU).   python scripts/r06/solo_phases.py [batches]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "32,256,512").split(",")]
B0, N, gates, kappa = sc.CONFIGS["headline"]
base = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B0)]
p0 = frx.Problem(base, sc.ZHANGJIAJIE, qd_intervals=kappa)
xs = p0.optimize(1e-6, x0=p0.initial_guess(), max_iterations=60)["x"]
for B in batches:
    rep = B // B0
    prob = frx.Problem(base * rep, sc.ZHANGJIAJIE, qd_intervals=kappa)
    xb = np.concatenate([xs] * rep)
    prob.set_eval_solo(2)
    rows = []
    for _ in range(5):
        st = np.zeros(32, np.int64)
        assert frx.lib().frx_profile_phases(prob.h, xb, st) == 0
        rows.append([int(st[13] - st[7]), int(st[14] - st[13]), int(st[15] - st[14]), int(st[15] - st[7])])
    r = np.median(np.array(rows), axis=0)
    print(json.dumps({"candidates": B, "cycles_forward": r[0], "cycles_penalty": r[1], "cycles_adjoint": r[2], "cycles_total": r[3], "us_at_2.4GHz": round(r[3] / 2400, 2),
                      "forward_phases": np.diff(st[:7]).tolist(), "adjoint_phases": np.diff(st[16:25]).tolist()}), flush=True)
    prob.close()
