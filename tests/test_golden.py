"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them (regression guard for the checker itself);
GPU: the HIP path reproduces them through the C ABI."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT

sys_path = os.path.join(ROOT, "tests", "golden")
FILES = sorted(f for f in glob.glob(os.path.join(sys_path, "*.npz")) if not os.path.basename(f).startswith(("corridor_", "refvs_", "mc512_")))   # corridor fixtures: test_next_rows.py; Monte-Carlo verdicts: test_gpu_configs.py + the CPU sample below


def load_case(path, sc):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(sys_path, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    name = os.path.splitext(os.path.basename(path))[0]
    sid, pid, N, gates, kappa, obst, over = mg.CASES[name]
    cand = sc.make_candidate(sid, N, gates, perturb_id=pid, obstacles=obst)
    return np.load(path), cand, kappa, over


def rel(a, b, floor=0.0):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), floor, 1e-300))


def test_fixtures_exist():
    assert len(FILES) >= 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(path, sc, ob):
    d, cand, kappa, over = load_case(path, sc)
    o = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=kappa, **over)
    assert rel(o.initial_guess(), d["x"][0]) < 1e-13
    for s, x in enumerate(d["x"]):
        f, g = o.objective(x)
        assert abs(f - d["f"][s]) <= 1e-12 * abs(d["f"][s])
        assert rel(g, d["g"][s], abs(f)) < 1e-10
        T, P, Cf = o.forward(x)
        assert rel(T, d["T"][s]) < 1e-14 and rel(Cf, d["C"][s]) < 1e-11
        c, gt, gc = o.penalty(T, Cf)
        assert abs(c - d["pen_cost"][s]) <= 1e-12 * max(abs(d["pen_cost"][s]), 1e-300)
    if "ref_f" in d:                                     # the reference's own outputs (oracle/_ref run at fixture time)
        assert rel(o.initial_guess(), d["ref_x0"]) < 1e-13
        for s, x in enumerate(d["x"]):
            f, g = o.objective(x)
            assert abs(f - d["ref_f"][s]) <= 1e-12 * abs(d["ref_f"][s]) and rel(g, d["ref_g"][s], abs(f)) < 1e-9
            T, P, Cf = o.forward(x)
            assert rel(Cf, d["ref_C"][s]) < 1e-10
    r = o.optimize(1e-6)
    # same compiler, same flags -> the very same iterate path
    assert r["iters"] == int(d["opt_iters"]) and r["status"] == int(d["opt_status"])
    assert rel(r["C"], d["opt_C"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_device_reproduces_golden(path, frx, sc):
    d, cand, kappa, over = load_case(path, sc)
    prob = frx.Problem([cand], sc.ZHANGJIAJIE, qd_intervals=kappa, **over)
    assert rel(prob.initial_guess(), d["x"][0]) < 1e-12
    for solver in ("knot_pcr", "banded_lu"):
        prob.set_solver(solver)
        for s, x in enumerate(d["x"]):
            f, g = prob.objective(x)
            assert abs(f[0] - d["f"][s]) <= 1e-9 * abs(d["f"][s])
            assert rel(g, d["g"][s], abs(d["f"][s])) < 1e-9
            T, Cf = prob.forward(x)
            assert rel(T, d["T"][s]) < 1e-13 and rel(Cf, d["C"][s]) < 1e-7
            hk = (np.repeat(d["T"][s], 6) ** np.tile(np.arange(6), len(d["T"][s])))[:, None]       # duration-normalised coefficients c_k h^k (what a coefficient
            assert rel(Cf * hk, d["C"][s] * hk) < 1e-11                                             # does to the trajectory): tests/test_gpu_parity.py, knot form
            cost, gdT, gdC = prob.penalty(d["T"][s], d["C"][s])
            assert abs(cost[0] - d["pen_cost"][s]) <= 1e-9 * max(abs(d["pen_cost"][s]), 1e-300)
            assert rel(gdT, d["pen_gdT"][s]) < 1e-9 and rel(gdC, d["pen_gdC"][s]) < 1e-9
            if "ref_f" in d:                             # against the REFERENCE's own numbers
                assert abs(f[0] - d["ref_f"][s]) <= 1e-9 * abs(d["ref_f"][s])
                assert rel(g, d["ref_g"][s], abs(d["ref_f"][s])) < 1e-9
                assert rel(Cf, d["ref_C"][s]) < 1e-7
                cost, gdT, gdC = prob.penalty(d["ref_T"][s], d["ref_C"][s])
                assert abs(cost[0] - d["ref_pen_cost"][s]) <= 1e-9 * max(abs(d["ref_pen_cost"][s]), 1e-300)
                assert rel(gdT, d["ref_pen_gdT"][s]) < 1e-9 and rel(gdC, d["ref_pen_gdC"][s]) < 1e-9
        # optimised coefficients: device map at the golden minimiser (the 1e-6 contract, lock-step form)
        T, Cf = prob.forward(d["opt_x"])
        assert rel(Cf, d["opt_C"]) < 1e-6 and rel(T, d["opt_T"]) < 1e-12
    prob.close()


# ---- the `Candidate` overload with the REFERENCE's vertices: a problem in the reference's own xi parameterisation ----
REFVS = ["refvs_n16_k8_obst", "refvs_n64_k16_obst"]          # the second: the benchmarked size (64 pieces x kappa 16) with obstacle planes, K_i = 8 ... 14


def _refvs_case(sc, name="refvs_n16_k8_obst"):
    d = np.load(os.path.join(sys_path, name + ".npz"))
    sid, pid, N, gates, kappa, obst = (int(v) for v in d["case"])
    cand = sc.make_candidate(sid, N, gates, perturb_id=pid, obstacles=bool(obst))
    vs = [d["v_rec"][3 * d["v_off"][m]:3 * d["v_off"][m + 1]].reshape(-1, 3).T.copy() for m in range(2 * N - 1)]
    assert int(d["polytopes_in_another_order_than_the_library"]) > 0             # the fixture is about a DIFFERENT vertex order (geoutils.hpp:43-149 + sdlp.hpp:689-708)
    return d, sc.Candidate(cand.ini_state, cand.fin_state, cand.h_polys, vs, cand.gates), kappa


@pytest.mark.parametrize("name", REFVS)
def test_oracle_and_host_setup_in_the_references_vertex_order(sc, ob, frx, name):
    d, cand, kappa = _refvs_case(sc, name)
    o = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=kappa)
    assert rel(o.initial_guess(), d["ref_x0"]) < 1e-12
    for s, x in enumerate(d["x"]):
        f, g = o.objective(x)
        assert abs(f - d["ref_f"][s]) <= 1e-12 * abs(d["ref_f"][s]) and rel(g, d["ref_g"][s], abs(f)) < 1e-9
    # the library's own host code (frx_host_setup.hpp, the code behind frx_initial_guess) on the same vertices
    from test_hostcheck import _host_guess, _load_hostcheck
    _, x0 = _host_guess(_load_hostcheck(), frx, [cand], sc.ZHANGJIAJIE, kappa, threads=1)
    assert rel(x0, d["ref_x0"]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", REFVS)
def test_device_plans_in_the_references_vertex_order(frx, sc, name):
    """frx_problem_create with the vertices geoutils::enumerateVs produced (fixture): initial guess, objective, gradient and coefficients equal the
    REFERENCE's own numbers in its parameterisation; a plan from there ends with the oracle's verdict at the oracle's objective level."""
    d, cand, kappa = _refvs_case(sc, name)
    prob = frx.Problem([cand], sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = prob.initial_guess()
    assert rel(x0, d["ref_x0"]) < 1e-12
    for s, x in enumerate(d["x"]):
        f, g = prob.objective(x)
        assert abs(f[0] - d["ref_f"][s]) <= 1e-9 * abs(d["ref_f"][s])
        assert rel(g, d["ref_g"][s], abs(d["ref_f"][s])) < 1e-9
        T, Cf = prob.forward(x)
        assert rel(T, d["ref_T"][s]) < 1e-13 and rel(Cf, d["ref_C"][s]) < 1e-7
    r = prob.optimize(1e-6, x0=x0)
    assert (r["status"][0] >= 0) == (int(d["opt_status"]) >= 0)
    assert abs(r["objective"][0] - float(d["opt_obj"])) <= 2e-2 * float(d["opt_obj"])       # independent runs of the reference's stop rule: DESIGN.md 4
    prob.close()


def test_monte_carlo_verdict_fixture_is_this_oracles_data(sc, ob):
    """tests/golden/mc512_cpu_verdicts.npz (make_mc_verdicts.py: four CPU variants of each of the 512 scenarios of one GPU's share of BASELINE configs[4]): a sample
    recomputed with the live oracle - status and objective of every variant - so that the fixture cannot drift from the checker it stands for (CPU, seconds)."""
    path = os.path.join(sys_path, "mc512_cpu_verdicts.npz")
    assert os.path.exists(path), "python tests/golden/make_mc_verdicts.py"
    fix = np.load(path)
    B, N, gates, kappa = sc.CONFIGS["montecarlo4096"]
    assert int(fix["first_id"]) == 0 and fix["status"].shape == (B // 8, 4) and int(fix["iteration_cap"]) == 60000
    for sid in (3, 400):
        cand = sc.make_candidate(sid, N, gates)
        for v, (mode, seed) in enumerate(fix["variants"]):
            o = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=kappa)
            o.set_abscissa_mode(bool(mode))
            x0 = o.initial_guess()
            if seed:
                x0 = x0 * (1.0 + 4e-16 * np.random.default_rng(int(seed)).integers(-2, 3, x0.size))
            r = o.optimize(sc.ZHANGJIAJIE["opt_rel_tol"], max_iterations=60000, x0=x0)
            assert int(r["status"]) == int(fix["status"][sid, v]) and int(r["iters"]) == int(fix["iters"][sid, v])
            assert abs(r["objective"] - fix["objective"][sid, v]) <= 1e-12 * abs(fix["objective"][sid, v])
    # the share's shape: every feasible scenario converges under every variant (LBFGS_STOP), the two infeasible ones fail under every variant
    st = fix["status"]
    assert int(np.sum(st.max(axis=1) < 0)) == 2 and int(np.sum((st.min(axis=1) < 0) & (st.max(axis=1) >= 0))) == 0
