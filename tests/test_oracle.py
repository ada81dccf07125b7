"""Self-checks that pin the CPU oracle: the reference holds no tests or golden
vectors for plan_manage, so the restatement is checked against mathematics it must satisfy."""
import numpy as np
import pytest


def dense_minco(T, head, tail):
    """Assemble the 6N x 6N MINCO matrix of SURVEY.md Appendix A.2 (CPU.hpp:437-499) densely, in numpy."""
    N = len(T); n = 6 * N
    A = np.zeros((n, n))
    A[0, 0] = 1; A[1, 1] = 1; A[2, 2] = 2
    for i in range(N - 1):
        t = T[i]; r = 6 * i
        A[r + 3, r + 3:r + 6] = [6, 24 * t, 60 * t ** 2]; A[r + 3, r + 9] = -6
        A[r + 4, r + 4:r + 6] = [24, 120 * t]; A[r + 4, r + 10] = -24
        A[r + 5, r:r + 6] = [1, t, t ** 2, t ** 3, t ** 4, t ** 5]
        A[r + 6, r:r + 6] = [1, t, t ** 2, t ** 3, t ** 4, t ** 5]; A[r + 6, r + 6] = -1
        A[r + 7, r + 1:r + 6] = [1, 2 * t, 3 * t ** 2, 4 * t ** 3, 5 * t ** 4]; A[r + 7, r + 7] = -1
        A[r + 8, r + 2:r + 6] = [2, 6 * t, 12 * t ** 2, 20 * t ** 3]; A[r + 8, r + 8] = -2
    t = T[-1]
    A[n - 3, n - 6:] = [1, t, t ** 2, t ** 3, t ** 4, t ** 5]
    A[n - 2, n - 5:] = [1, 2 * t, 3 * t ** 2, 4 * t ** 3, 5 * t ** 4]
    A[n - 1, n - 4:] = [2, 6 * t, 12 * t ** 2, 20 * t ** 3]
    return A


@pytest.fixture(scope="module")
def small(sc, ob):
    c = sc.make_candidate(2, 16, 4, obstacles=True)
    o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=8)
    return c, o


def test_banded_lu_solve_and_adjoint_match_dense(small, sc):
    c, o = small
    x = o.initial_guess()
    T, P, Cf = o.forward(x)
    N = o.fine_n
    A = dense_minco(T, c.ini_state, c.fin_state)
    # L*U of the banded factorisation reproduces A
    assert np.abs(o.dense_A(T) - A).max() <= 1e-12 * np.abs(A).max()
    rhs = np.zeros((6 * N, 3))
    rhs[0:3] = c.ini_state.T; rhs[6 * N - 3:] = c.fin_state.T
    for i in range(N - 1):
        rhs[6 * i + 5] = P[i]
    ref = np.linalg.solve(A, rhs)
    assert np.abs(Cf - ref).max() <= 1e-9 * np.abs(ref).max()
    rng = np.random.default_rng(0)
    r = rng.standard_normal((6 * N, 3))
    assert np.abs(o.solve_adj(r) - np.linalg.solve(A.T, r)).max() <= 1e-9 * np.abs(r).max() * np.linalg.cond(A) * 1e-6 + 1e-9
    assert np.abs(o.solve(r) - np.linalg.solve(A, r)).max() <= 1e-8 * np.abs(np.linalg.solve(A, r)).max()


def poly_eval(c6x3, t, d=0):
    k = np.arange(6)
    coef = np.ones(6)
    for j in range(d):
        coef = coef * (k - j)
    pw = np.where(k - d >= 0, t ** np.maximum(k - d, 0), 0.0)
    return (coef * pw) @ c6x3


def test_spline_invariants(small):
    c, o = small
    x = o.initial_guess()
    rng = np.random.default_rng(1)
    x = x + 0.1 * rng.standard_normal(x.size)
    T, P, Cf = o.forward(x)
    N = o.fine_n
    pc = Cf.reshape(N, 6, 3)
    scale = np.abs(Cf).max()
    for d in range(3):                                  # head / tail PVA reproduced
        assert np.abs(poly_eval(pc[0], 0.0, d) - c.ini_state[:, d]).max() < 1e-9 * scale
        assert np.abs(poly_eval(pc[-1], T[-1], d) - c.fin_state[:, d]).max() < 1e-7 * scale
    for i in range(N - 1):
        assert np.abs(poly_eval(pc[i], T[i], 0) - P[i]).max() < 1e-8 * scale      # waypoint interpolated
        for d in range(5):                               # p, v, a, j, s continuous at the junction
            a, b = poly_eval(pc[i], T[i], d), poly_eval(pc[i + 1], 0.0, d)
            assert np.abs(a - b).max() < 1e-6 * max(np.abs(a).max(), 1.0), (i, d)


def test_jerk_cost_closed_form_vs_quadrature(small):
    c, o = small
    x = o.initial_guess()
    T, P, Cf = o.forward(x)
    pc = Cf.reshape(o.fine_n, 6, 3)
    gx, gw = np.polynomial.legendre.leggauss(8)
    J = 0.0
    for i in range(o.fine_n):
        for xq, wq in zip(gx, gw):
            t = 0.5 * T[i] * (xq + 1)
            j = poly_eval(pc[i], t, 3)
            J += 0.5 * T[i] * wq * (j @ j)
    assert abs(o.jerk_cost() - J) <= 1e-10 * J


def test_diffeomorphisms_invert(small):
    c, o = small
    rng = np.random.default_rng(2)
    x = o.initial_guess()
    x = x + 0.2 * rng.standard_normal(x.size)
    T, P, _ = o.forward(x)
    xb = o.backward(T, P)                                # one piece per polytope: coarse T = fine T
    T2, P2, _ = o.forward(xb)
    assert np.abs(T2 - T).max() <= 1e-12 * T.max()
    assert np.abs(P2 - P).max() <= 1e-5                  # NLS stops at g_epsilon = FLT_EPSILON (CPU.hpp:789)


@pytest.mark.parametrize("kappa", [8, 16])
def test_gradient_matches_finite_differences(sc, ob, kappa):
    c = sc.make_candidate(4, 16, 4, obstacles=True)
    o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = o.initial_guess()
    pts = [x0, o.optimize(1e-6, max_iterations=30, x0=x0)["x"], o.optimize(1e-6, max_iterations=200, x0=x0)["x"]]
    rng = np.random.default_rng(3)
    for x in pts:
        f, g = o.objective(x)
        for _ in range(3):
            d = rng.standard_normal(o.n); d /= np.linalg.norm(d)
            best = np.inf
            for h in (1e-4, 1e-5, 1e-6, 1e-7):          # 2nd/4th-order central differences; keep the best step (round-off vs truncation)
                fp, fm = o.objective(x + h * d)[0], o.objective(x - h * d)[0]
                fp2, fm2 = o.objective(x + 2 * h * d)[0], o.objective(x - 2 * h * d)[0]
                for fd in ((fp - fm) / (2 * h), (-fp2 + 8 * fp - 8 * fm + fm2) / (12 * h)):
                    best = min(best, abs(fd - g @ d) / max(abs(g @ d), 1e-12))
            assert best < 5e-6, best


def test_penalty_vanishes_inside_corridor_and_limits(sc, ob):
    """A slow trajectory strictly inside corridor, speed, thrust and body-rate limits has zero penalty
    and a jerk-only gradient (SURVEY.md §8c check 6)."""
    c = sc.make_candidate(5, 8, 2)
    o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=16)
    Tc, P = o.set_initial()
    x = o.backward(Tc * 6.0, P)                          # six times slower than the 10 m/s allocation
    T, P2, Cf = o.forward(x)
    cost, gdT, gdC = o.penalty(T, Cf)
    assert cost == 0.0 and not gdT.any() and not gdC.any()
    f, g = o.objective(x)
    assert abs(f - (o.jerk_cost() + sc.ZHANGJIAJIE["rho"] * T.sum())) <= 1e-12 * f


def test_abscissa_modes_agree_to_rounding(small):
    c, o = small
    x = o.optimize(1e-6, max_iterations=40)["x"]
    o.set_abscissa_mode(True); f1, g1 = o.objective(x)
    o.set_abscissa_mode(False); f2, g2 = o.objective(x)
    o.set_abscissa_mode(True)
    assert abs(f1 - f2) <= 1e-9 * abs(f1) and np.abs(g1 - g2).max() <= 1e-9 * np.abs(g1).max()


def test_intervals_greater_than_one(sc, ob):
    """gridRes < inf splits polytopes into several pieces (CPU.hpp:1003-1029, 1129-1152, 930-959)."""
    c = sc.make_candidate(6, 10, 2)
    o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=8, grid_res=1.5)
    iv, ivs, ihs = o.maps()
    assert o.fine_n == iv.sum() and o.fine_n > o.coarse_n and o.dim_t == o.coarse_n
    assert list(ihs) == [i for i, k in enumerate(iv) for _ in range(k)]
    x = o.initial_guess()
    f, g = o.objective(x)
    rng = np.random.default_rng(4)
    d = rng.standard_normal(o.n); d /= np.linalg.norm(d)
    errs = [abs((o.objective(x + h * d)[0] - o.objective(x - h * d)[0]) / (2 * h) - g @ d) / abs(g @ d) for h in (1e-5, 1e-6, 1e-7)]
    assert min(errs) < 2e-6


def test_optimize_reaches_feasible_fast_trajectory(sc, ob):
    c = sc.make_candidate(0, 32, 8)
    o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=8)
    r = o.optimize(1e-6)
    assert r["status"] in (0, 1)
    f0 = o.objective(o.initial_guess())[0]
    assert r["objective"] < 1e-6 * f0
    pen, _, _ = o.penalty(r["T"], r["C"])
    assert pen < 1e-2 * r["objective"]                    # constraints met up to the soft-penalty residual
    assert 5.0 < r["T"].sum() < 30.0
