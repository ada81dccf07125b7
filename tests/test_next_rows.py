"""SURVEY.md §8f "next" rows built so far: f1 (H -> V vertex enumeration) and f3 (result wire format + its consumer)."""
import numpy as np
import pytest


def test_vertex_enumeration_matches_generator_and_reference(frx, sc, ob):
    cand = sc.make_candidate(4, 12, 3, obstacles=True)
    ref = ob.Reference(cand, sc.ZHANGJIAJIE, override_vs=False, qd_intervals=8) if ob.ref_gcopter() is not None else None
    for m, V in enumerate(cand.v_polys):
        i = m // 2
        H = cand.h_polys[i] if m % 2 == 0 else np.concatenate([cand.h_polys[i], cand.h_polys[i + 1]], axis=1)
        Vl = frx.enumerate_vertices(H)
        assert Vl.shape == V.shape and np.abs(Vl - V).max() < 1e-9          # same algorithm as the generator: same order
        kq = np.round(Vl / 1e-7)
        assert np.all(np.lexsort((kq[2], kq[1], kq[0])) == np.arange(Vl.shape[1]))
        if ref is not None:                                                  # geoutils::enumerateVs: same vertex SET
            Vr = ref.vpoly(m)
            assert Vr.shape == Vl.shape
            d = np.abs(Vl.T[:, None, :] - Vr.T[None, :, :]).max(axis=2).min(axis=1)
            assert d.max() < 1e-6


def test_empty_polytope_is_reported(frx):
    import ctypes as C
    # two opposing half-spaces that exclude each other
    H = np.array([[1, 0, 0, -1, 0, 0], [-1, 0, 0, 1, 0, 0], [0, 1, 0, 0, 1, 0], [0, -1, 0, 0, -1, 0], [0, 0, 1, 0, 0, 1], [0, 0, -1, 0, 0, -1]], float).T
    nv = C.c_int()
    rc = frx.lib().frx_enumerate_vertices(6, np.ascontiguousarray(H.T.reshape(-1)), None, 0, C.byref(nv))
    assert rc == -4                                                          # FRX_ERR_EMPTY_POLYTOPE


def test_wire_format_round_trip(frx, sc, ob):
    """traj2msg fields + traj_server sampling reproduce the optimised polynomials (position .. jerk) at arbitrary times."""
    cand = sc.make_candidate(2, 10, 2)
    o = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=8)
    r = o.optimize(1e-6, max_iterations=60)
    T, Cf = r["T"], r["C"]
    msg = frx.traj_to_msg(T, Cf)
    cx, cy, cz, tm, od = msg
    assert np.array_equal(tm, T) and np.all(od == 5)
    pc = Cf.reshape(-1, 6, 3)
    # coefficient layout: piece i, column j <-> power 5-j, scaled by T^(5-j)
    for i in (0, 4, 9):
        for j in range(6):
            assert cx[6 * i + j] == pytest.approx(pc[i, 5 - j, 0] * T[i] ** (5 - j), rel=1e-14, abs=1e-300)
    rng = np.random.default_rng(0)
    edges = np.concatenate([[0.0], np.cumsum(T)])
    for t in list(rng.uniform(0, edges[-1], 40)) + [0.0, float(edges[3]), float(edges[-1]), float(edges[-1]) + 1.0]:
        p, v, a, j = frx.msg_sample(msg, t)
        i = min(int(np.searchsorted(edges, t, side="left")) - 1, len(T) - 1) if t > 0 else 0
        i = max(i, 0)
        tl = t - edges[i]
        k = np.arange(6)
        def ev(dn):
            coef = np.ones(6)
            for q in range(dn):
                coef = coef * (k - q)
            return (coef * np.where(k - dn >= 0, tl ** np.maximum(k - dn, 0), 0.0)) @ pc[i]
        for got, dn in ((p, 0), (v, 1), (a, 2), (j, 3)):
            want = ev(dn)
            assert np.abs(got - want).max() <= 1e-9 * max(np.abs(want).max(), 1.0), (t, dn)


@pytest.mark.gpu
def test_create_from_h_equals_create_with_supplied_vertices(frx, sc):
    cands = sc.make_batch(3, 3, 16, 4, obstacles=True)
    a = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=8)
    b = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=8, enumerate_v=True)
    assert np.array_equal(a.x_off, b.x_off)
    xa, xb = a.initial_guess(), b.initial_guess()
    assert np.abs(xa - xb).max() < 1e-9
    fa, ga = a.objective(xa); fb, gb = b.objective(xa)
    assert np.all(np.abs(fa - fb) <= 1e-9 * np.abs(fa)) and np.abs(ga - gb).max() <= 1e-9 * np.abs(ga).max()
    a.close(); b.close()
