"""Cycle stamps of cluster 0 during one evaluation in the one-launch form (frx_eval_kernel.hpp), relative to the leader's entry, in microseconds at the shader clock
the device reports.   python scripts/r05/eval_fused_timeline.py [config]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from frx_import import frx
from fast_racing_amd import scenario as sc
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
B, N, gates, kappa = sc.CONFIGS[name]
cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
xs = prob.optimize(sc.ZHANGJIAJIE["opt_rel_tol"], x0=prob.initial_guess(), max_iterations=60)["x"]
mhz = torch.cuda.get_device_properties(0).clock_rate / 1e3 if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 2400.0
runs = []
for rep in range(7):
    st = prob.profile_eval_cluster(xs).astype(np.float64)
    r = {}
    for i in range(64):
        if st[i] == 0 or i == 49: continue
        if 40 <= i <= 48: r[i] = (st[i] - st[40]) / 100.0                       # 100 MHz ticks -> us
        else: r[i] = (st[i] - st[49])                                          # shader cycles since the leader's entry
    runs.append(r)
keys = sorted(set.intersection(*[set(r) for r in runs]))
med = {k: float(np.median([r[k] for r in runs])) for k in keys}
# shader clock: the leader's adjoint ends (stamp 23, cycles) just before stamp 42 (us)
ghz = med[23] / med[42] / 1e3 if 23 in med and 42 in med and med[42] > 0 else 2.4
names = {0: "fwd entry", 1: "fwd staged", 2: "fwd durations", 5: "fwd matrix wave done", 6: "fwd C swept", 8: "fwd ax waypoints", 9: "fwd ax barrier", 10: "fwd ax rhs", 11: "fwd ax pcr done", 12: "fwd ax hermite",
         16: "bwd entry", 17: "bwd loads", 22: "bwd barrier", 23: "bwd end", 24: "bwd tap end", 25: "bwd ax start", 26: "bwd ax poll + hermite adj", 27: "bwd ax solve", 28: "bwd ax knot adj", 29: "bwd ax end", 30: "bwd ax wp a", 31: "bwd ax wp b",
         40: "LEADER entry", 41: "LEADER fwd done", 42: "LEADER bwd done", 43: "LEADER end", 44: "MEMBER entry", 45: "MEMBER gate seen", 46: "MEMBER granules staged", 47: "MEMBER samples done", 48: "MEMBER partials out"}
us = {k: (med[k] if 40 <= k <= 48 else med[k] / (ghz * 1e3)) for k in med}
print(json.dumps({"config": name, "shader_ghz_estimated": round(ghz, 3), "us_since_leader_entry": {f"{k}:{names.get(k, '')}": round(us[k], 2) for k in sorted(us, key=lambda q: us[q])}}, indent=1))
