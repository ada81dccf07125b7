import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B, N, gates, kappa = sc.CONFIGS["headline"]
prob = frx.Problem([sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)], sc.ZHANGJIAJIE, qd_intervals=kappa)
x = prob.optimize(1e-6, max_iterations=60)["x"]
for rep in range(3):
    st = np.zeros(32, np.int64)
    assert frx.lib().frx_profile_phases(prob.h, x, st) == 0
    f = st[:7]; bwd = st[16:25]
    print("fwd phase cycles:", np.diff(f), "total", f[-1]-f[0], "| bwd:", np.diff(bwd), "total", bwd[-1]-bwd[0])
