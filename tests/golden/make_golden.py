"""Generates tests/golden/*.npz: small seeded problems with the CPU oracle's outputs, so that
  - the oracle is guarded against silent regressions (tests/test_golden.py, CPU),
  - the GPU path is compared against committed numbers as well as against the live oracle.
The reference has no fixtures of its own for this path.  The `ref_*` arrays are OUTPUTS OF THE REFERENCE
ITSELF: its CPU path (se3gcopter_cpu.hpp, trajectory.hpp, lbfgs.hpp, ...) compiled unmodified from /root/reference against
oracle/eigen_shim (oracle/_ref/libref_gcopter.so), run on these inputs in this container.  The other arrays come from the
restatement in oracle/gcopter_oracle.cpp (which tests/test_reference_pin.py pins to the same library).  /root/reference
does not exist on the GPU box, which is why the vectors are committed.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frx_import import frx  # noqa: E402,F401
from fast_racing_amd import scenario as sc  # noqa: E402
from oracle import binding as ob  # noqa: E402

CASES = {
    # name: (scenario_id, perturb_id, N pieces, gates, kappa, obstacles, overrides)
    "n8_k8": (21, 0, 8, 2, 8, False, {}),
    "n16_k16_obst": (22, 1, 16, 4, 16, True, {}),
    "n12_k48_stock": (23, 0, 12, 3, 48, False, {}),
    "n10_fixedT_exp": (24, 2, 10, 2, 8, False, dict(rho=0.0, total_t=6.0, c2_diffeo=0)),
    # the size every number of the bench line is quoted on (BASELINE.json configs[2]: 64 pieces x kappa 16): candidate 0 of the bench batch
    # (K_i = 8) and a perturbed candidate of the same scenario with obstacle planes (K_i = 8 ... 14)
    "headline_n64_k16": (0, 0, 64, 16, 16, False, {}),
    "n64_k16_obst": (0, 3, 64, 16, 16, True, {}),
}


def build(name):
    sid, pid, N, gates, kappa, obst, over = CASES[name]
    cand = sc.make_candidate(sid, N, gates, perturb_id=pid, obstacles=obst)
    o = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=kappa, **over)
    x0 = o.initial_guess()
    xs = [x0, o.optimize(1e-6, max_iterations=10, x0=x0)["x"], o.optimize(1e-6, max_iterations=80, x0=x0)["x"]]
    out = dict(case=np.array([sid, pid, N, gates, kappa, int(obst)]), x=np.array(xs))
    fs, gs, Ts, Cs, pc, pt, pg = [], [], [], [], [], [], []
    for x in xs:
        f, g = o.objective(x); fs.append(f); gs.append(g)
        T, P, Cf = o.forward(x); Ts.append(T); Cs.append(Cf)
        c, gt, gc = o.penalty(T, Cf); pc.append(c); pt.append(gt); pg.append(gc)
    if ob.ref_gcopter() is not None:
        R = ob.Reference(cand, sc.ZHANGJIAJIE, override_vs=True, qd_intervals=kappa, **over)
        rf, rg, rT, rC, rpc, rpt, rpg = [], [], [], [], [], [], []
        for x in xs:
            f, g = R.objective(x); rf.append(f); rg.append(g)
            T, Cf = R.forward(x); rT.append(T); rC.append(Cf)
            c, gt, gc = R.penalty(T, Cf); rpc.append(c); rpt.append(gt); rpg.append(gc)
        out.update(ref_x0=R.initial_guess(), ref_f=np.array(rf), ref_g=np.array(rg), ref_T=np.array(rT), ref_C=np.array(rC),
                   ref_pen_cost=np.array(rpc), ref_pen_gdT=np.array(rpt), ref_pen_gdC=np.array(rpg))
    r = o.optimize(1e-6)
    out.update(f=np.array(fs), g=np.array(gs), T=np.array(Ts), C=np.array(Cs), pen_cost=np.array(pc), pen_gdT=np.array(pt),
               pen_gdC=np.array(pg), opt_x=r["x"], opt_C=r["C"], opt_T=r["T"], opt_obj=np.array(r["objective"]),
               opt_jerk=np.array(r["jerk_cost"]), opt_iters=np.array(r["iters"]), opt_evals=np.array(r["evals"]),
               opt_status=np.array(r["status"]))
    return out


# the reference's OWN vertex order and outputs: a small case, and the benchmarked size with obstacle planes (64 pieces x kappa 16, K_i = 8 ... 14)
REFVS_CASES = {"refvs_n16_k8_obst": (25, 0, 16, 4, 8), "refvs_n64_k16_obst": (0, 3, 64, 16, 16)}


def build_refvs(sid=25, pid=0, N=16, gates=4, kappa=8):
    """The `Candidate` overload of INTEGRATION.md 2: the caller keeps geoutils::enumerateVs on the reference side and hands the vertices over.
    The V-polytopes here are the REFERENCE's own (its Seidel LP + quickhull order, geoutils.hpp:43-149 - NOT the lexicographic order of
    frx_enumerate_vertices), so x lives in the reference's xi parameterisation; ref_* are the reference's outputs in that parameterisation."""
    cand = sc.make_candidate(sid, N, gates, perturb_id=pid, obstacles=True)
    R = ob.Reference(cand, sc.ZHANGJIAJIE, override_vs=False, qd_intervals=kappa)          # its own enumerateVs
    vs = [R.vpoly(m) for m in range(2 * N - 1)]
    cand_r = sc.Candidate(cand.ini_state, cand.fin_state, cand.h_polys, vs, cand.gates)
    o = ob.Oracle(cand_r, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = R.initial_guess()
    xs = [x0, o.optimize(1e-6, max_iterations=10, x0=x0)["x"], o.optimize(1e-6, max_iterations=120, x0=x0)["x"]]
    rf, rg, rT, rC = [], [], [], []
    for x in xs:
        f, g = R.objective(x); rf.append(f); rg.append(g)
        T, Cf = R.forward(x); rT.append(T); rC.append(Cf)
    ro = o.optimize(1e-6, x0=x0)
    v_off = np.cumsum([0] + [v.shape[1] for v in vs]).astype(np.int32)
    differs = sum(int(v.shape != w.shape or np.abs(v - w).max() > 1e-6) for v, w in zip(vs, cand.v_polys))
    return dict(case=np.array([sid, pid, N, gates, kappa, 1]), v_off=v_off, v_rec=np.concatenate([v.T.reshape(-1) for v in vs]), x=np.array(xs),
                ref_x0=x0, ref_f=np.array(rf), ref_g=np.array(rg), ref_T=np.array(rT), ref_C=np.array(rC),
                opt_obj=np.array(ro["objective"]), opt_status=np.array(ro["status"]), opt_iters=np.array(ro["iters"]),
                polytopes_in_another_order_than_the_library=np.array(differs))


def build_corridor(seed):
    """Corridor fixture: cells by the REFERENCE's decomp_util (oracle/_ref/libref_decomp.so) inside the oracle's restatement
    of the corridor loop of MavGlobalPlanner::plan, on a synthetic front-end path and point cloud."""
    rng = np.random.default_rng(seed)
    gates = sc.make_gates(sc.SplitMix64(seed), 4)
    wps = np.vstack([[0.0, 0.0, 1.0], gates, gates[-1] + [0.0, 15.0, 0.0]])
    path = [wps[0]]
    for a, b in zip(wps[:-1], wps[1:]):
        m = int(np.ceil(np.linalg.norm(b - a) / 0.5))
        path += [a + (b - a) * (t / m) for t in range(1, m + 1)]
    path = np.array(path)
    obs = []
    while len(obs) < 500:
        q = path[rng.integers(len(path))] + rng.normal(0, 3.0, 3)
        if 0.0 < q[2] < 3.0 and np.min(np.linalg.norm(path - q, axis=1)) > 0.6:
            obs.append(q)
    obs = np.array(obs); bbox = np.array([4.0, 4.0, 2.5])
    polys = ob.corridor_oracle(path, obs, bbox, 3.0)
    h_off = np.cumsum([0] + [h.shape[1] for h in polys]).astype(np.int32)
    return dict(path=path, obs=obs, bbox=bbox, map_height=np.array(3.0), h_off=h_off, h_rec=np.concatenate([h.T.reshape(-1) for h in polys]))


if __name__ == "__main__":
    if ob.ref_decomp() is not None:
        d = build_corridor(5)
        np.savez_compressed(os.path.join(os.path.dirname(__file__), "corridor_seed5.npz"), **d)
        print("corridor_seed5:", len(d["h_off"]) - 1, "cells,", int(d["h_off"][-1]), "half-spaces")
    if "--corridor-only" in sys.argv:
        sys.exit(0)
    if ob.ref_gcopter() is not None:
        for name, args in REFVS_CASES.items():
            if "--refvs-new-only" in sys.argv and os.path.exists(os.path.join(os.path.dirname(__file__), name + ".npz")): continue
            d = build_refvs(*args)
            np.savez_compressed(os.path.join(os.path.dirname(__file__), name + ".npz"), **d)
            print(name, ": f =", d["ref_f"], "polytopes whose vertex order differs from frx_enumerate_vertices:", int(d["polytopes_in_another_order_than_the_library"]), "of", len(d["v_off"]) - 1)
    if "--refvs-new-only" in sys.argv:
        sys.exit(0)
    for name in CASES:
        d = build(name)
        np.savez_compressed(os.path.join(os.path.dirname(__file__), name + ".npz"), **d)
        print(name, "f =", d["f"], "opt obj", float(d["opt_obj"]), "iters", int(d["opt_iters"]))
