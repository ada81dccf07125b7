"""How often does the cheap half-space pre-reject  d0 < -max(ell)  hold?  (CPU, oracle trajectories.)
Per (sample, half-space), per (lane, chunk of 4) and per (wave of 3 pieces x 17 samples, chunk of 4): the wave-level figure is what a
wave-uniform branch can skip.  Usage: python scripts/r05/prereject_stats.py [n_cands] [iterations...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from frx_import import frx  # noqa: F401  (package path with the hyphen)
from fast_racing_amd import scenario as sc
from oracle import binding as ob

ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = [int(a) for a in sys.argv[2:]] or [0, 60, 0x7fffffff]
B, N, gates, kappa = sc.CONFIGS["headline"]
P = sc.ZHANGJIAJIE
emax = max(P["horiz_half_len"], P["vert_half_len"])
for it in iters:
    tot = dict(pairs=0, pairs_rej=0, lane_chunks=0, lane_chunks_rej=0, wave_chunks=0, wave_chunks_rej=0, wave_all=0, wave_all_rej=0, viol=0)
    for b in range(ncand):
        c = sc.make_candidate(0, N, gates, perturb_id=b)
        o = ob.Oracle(c, P, qd_intervals=kappa)
        x0 = o.initial_guess()
        if it == 0: x = x0
        else: x = o.optimize(P["opt_rel_tol"], max_iterations=(0 if it == 0x7fffffff else it), x0=x0)["x"]
        T, _, Cf = o.forward(x)
        Cf = Cf.reshape(N, 6, 3)
        rej = np.zeros((N, kappa + 1, 8), bool)
        for i in range(N):
            H = c.h_polys[i]                       # 6 x K: (outer normal, point)
            n = H[:3] / np.linalg.norm(H[:3], axis=0); p = H[3:]
            s = np.arange(kappa + 1) * (T[i] / kappa)
            pw = np.stack([s ** k for k in range(6)], 1)      # [S][6]
            pos = pw @ Cf[i]                                   # [S][3]
            d0 = (pos[:, None, :] - p.T[None]) @ np.ones(3) * 0  # placeholder
            d0 = np.einsum('skd,dk->sk', pos[:, None, :] - p.T[None], n) + P["safe_margin"]
            K = H.shape[1]
            rej[i, :, :K] = d0 < -emax * (1 + 1e-9)
            rej[i, :, K:] = True
            tot["viol"] += int((d0 > -0.0).sum())
        tot["pairs"] += rej.size; tot["pairs_rej"] += int(rej.sum())
        lc = rej.reshape(N, kappa + 1, 2, 4).all(-1)
        tot["lane_chunks"] += lc.size; tot["lane_chunks_rej"] += int(lc.sum())
        for w0 in range(0, N, 3):
            wv = lc[w0:w0 + 3].all((0, 1))
            tot["wave_chunks"] += wv.size; tot["wave_chunks_rej"] += int(wv.sum())
            tot["wave_all"] += 1; tot["wave_all_rej"] += int(wv.all())
    print("iterations", "converged" if it == 0x7fffffff else it, {k: v for k, v in tot.items()},
          "pair %.3f lane-chunk %.3f wave-chunk %.3f wave-all %.3f" % (tot["pairs_rej"] / tot["pairs"], tot["lane_chunks_rej"] / tot["lane_chunks"],
                                                                     tot["wave_chunks_rej"] / tot["wave_chunks"], tot["wave_all_rej"] / tot["wave_all"]))
