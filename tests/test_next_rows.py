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


# ---------------------------------------------------------------------------------------------------------------------
# f2: corridor generation
# ---------------------------------------------------------------------------------------------------------------------
def _canon(H):
    """half-spaces in a canonical order: the reference's order among planes tangent at points that all lie ON the final
    ellipsoid (distance 1 +- 1 ulp each, e.g. the two contact points that fixed its short axes) is decided by rounding"""
    key = np.round(H / 1e-6).astype(np.int64)
    return H[:, np.lexsort(key[::-1])]


def _cloud(rng, p1, p2, n, spread=3.0, clear=0.35):
    """random obstacle points around segment p1-p2, none closer than `clear` to the segment"""
    pts = []
    d = p2 - p1; L = np.linalg.norm(d); u = d / L
    while len(pts) < n:
        q = p1 + u * rng.uniform(-2.0, L + 2.0) + rng.normal(0, spread, 3)
        t = np.clip((q - p1) @ u, 0, L)
        if np.linalg.norm(q - (p1 + t * u)) > clear:
            pts.append(q)
    return np.array(pts)


@pytest.mark.parametrize("seed", range(12))
def test_line_segment_cell_matches_reference_decomp(frx, ob, seed):
    """frx_line_segment_dilate vs the reference's LineSegment3D::dilate (compiled from its own headers): same half-spaces in
    the same order, same ellipsoid."""
    if ob.ref_decomp() is None:
        pytest.skip("oracle/_ref/libref_decomp.so not built (reference tree absent)")
    rng = np.random.default_rng(100 + seed)
    p1 = rng.uniform(-5, 5, 3); p2 = p1 + rng.normal(0, 1, 3) * np.array([3.0, 3.0, 0.6])
    if seed == 3: p2 = p1 + np.array([0.0, 0.0, 2.0])            # vertical segment: the degenerate branch of add_local_bbox
    n = [0, 1, 5, 40, 200, 1000][seed % 6]
    obs = _cloud(rng, p1, p2, n) if n else np.zeros((0, 3))
    bbox = np.array([4.0, 4.0, 2.5]) if seed != 7 else np.zeros(3)
    H, Cm, d = frx.line_segment_dilate(p1, p2, bbox, obs)
    Hr, Cr, dr = ob.ref_line_segment_dilate(p1, p2, bbox, obs)
    assert H.shape == Hr.shape
    assert np.abs(_canon(H) - _canon(Hr)).max() < 1e-9 and np.abs(Cm - Cr).max() < 1e-9 and np.abs(d - dr).max() < 1e-12
    # every obstacle inside the local box is outside (or on) some tangent plane; the segment itself is inside the cell
    for q in (p1, p2, 0.5 * (p1 + p2)):
        assert np.all(np.einsum("dk,dk->k", H[:3], q[:, None] - H[3:]) <= 1e-9)


def _scene(sc, seed, n_gates=4, n_obs=600):
    rng = np.random.default_rng(seed)
    g = sc.SplitMix64(seed)
    gates = sc.make_gates(g, n_gates)
    wps = np.vstack([[0.0, 0.0, 1.0], gates, gates[-1] + [0.0, 15.0, 0.0]])
    path = [wps[0]]
    for a, b in zip(wps[:-1], wps[1:]):                           # dense front-end-like path, 0.5 m spacing
        m = int(np.ceil(np.linalg.norm(b - a) / 0.5))
        path += [a + (b - a) * (t / m) for t in range(1, m + 1)]
    path = np.array(path)
    obs = []
    while len(obs) < n_obs:
        q = path[rng.integers(len(path))] + rng.normal(0, 3.0, 3)
        if 0.0 < q[2] < 3.0 and np.min(np.linalg.norm(path - q, axis=1)) > 0.6:
            obs.append(q)
    return path, np.array(obs)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_corridor_matches_oracle_loop_around_reference_cells(frx, sc, ob, seed):
    if ob.ref_decomp() is None:
        pytest.skip("oracle/_ref/libref_decomp.so not built (reference tree absent)")
    path, obs = _scene(sc, seed)
    bbox = np.array(sc.ZHANGJIAJIE.get("polyhedron_box", [4.0, 4.0, 2.5]), dtype=float)
    blocked = (lambda a, b: bool(np.linalg.norm(a - b) > 3.0 and (int(a[1]) + int(b[1])) % 7 == 0)) if seed == 3 else None
    got = frx.corridor_generate(path, obs, bbox, 3.0, blocked=blocked)
    want = ob.corridor_oracle(path, obs, bbox, 3.0, blocked=blocked)
    assert len(got) == len(want) and len(got) >= 4
    for H, Hw in zip(got, want):
        assert H.shape == Hw.shape and np.abs(_canon(H) - _canon(Hw)).max() < 1e-9
    # the corridor is usable: consecutive cells overlap with interior, and the whole path is covered
    for H0, H1 in zip(got[:-1], got[1:]):
        assert frx.enumerate_vertices(np.concatenate([H0, H1], axis=1)).shape[1] >= 4
    for q in path:
        assert any(np.all(np.einsum("dk,dk->k", H[:3], q[:, None] - H[3:]) <= 1e-9) for H in got)


def test_corridor_golden(frx):
    """committed fixture made with the reference's decomp_util (tests/golden/make_golden.py): runs without the reference tree"""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "corridor_seed5.npz"))
    got = frx.corridor_generate(z["path"], z["obs"], z["bbox"], float(z["map_height"]))
    assert len(got) == len(z["h_off"]) - 1
    for k, H in enumerate(got):
        Hw = z["h_rec"][6 * z["h_off"][k]:6 * z["h_off"][k + 1]].reshape(-1, 6).T
        assert H.shape == Hw.shape and np.abs(_canon(H) - _canon(Hw)).max() < 1e-9


@pytest.mark.gpu
def test_path_to_plan_through_the_library(frx, sc):
    """front-end path + point cloud -> corridor (f2) -> vertices (f1) -> optimiser -> message (f3), all through the C ABI"""
    path, obs = _scene(sc, 11, n_gates=3, n_obs=400)
    polys = frx.corridor_generate(path, obs, np.array([4.0, 4.0, 2.5]), 3.0)
    ini = np.zeros((3, 3)); ini[:, 0] = path[0]
    fin = np.zeros((3, 3)); fin[:, 0] = path[-1]
    cand = sc.Candidate(ini_state=ini, fin_state=fin, h_polys=polys, v_polys=[], gates=np.zeros((0, 3)))
    prob = frx.Problem([cand], sc.ZHANGJIAJIE, qd_intervals=8, enumerate_v=True)
    r = prob.optimize(1e-5, max_iterations=400)
    assert r["status"][0] >= 0 or r["iters"][0] == 400
    T = r["T"][:prob.P]; Cf = r["C"][:6 * prob.P]
    msg = frx.traj_to_msg(T, Cf)
    p0, v0, _, _ = frx.msg_sample(msg, 0.0)
    pe, ve, _, _ = frx.msg_sample(msg, float(T.sum()))
    assert np.abs(p0 - path[0]).max() < 1e-6 and np.abs(pe - path[-1]).max() < 1e-6 and np.abs(v0).max() < 1e-6 and np.abs(ve).max() < 1e-6
    # the optimised trajectory stays inside the corridor it was given (sampled; margin of the penalty's soft constraint)
    edges = np.concatenate([[0.0], np.cumsum(T)])
    for t in np.linspace(0, edges[-1], 200):
        p, _, _, _ = frx.msg_sample(msg, float(t))
        assert any(np.all(np.einsum("dk,dk->k", H[:3], p[:, None] - H[3:]) <= 0.3) for H in polys)
    prob.close()


# ---------------------------------------------------------------------------------------------------------------------
# f3 against the reference's own Piece / Trajectory / RootFinder
# ---------------------------------------------------------------------------------------------------------------------
def _optimised(sc, ob, sid=2, N=10, gates=2):
    cand = sc.make_candidate(sid, N, gates)
    r = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=8).optimize(1e-6, max_iterations=80)
    return r["T"], r["C"]


def test_max_rates_match_reference_root_finder(frx, sc, ob):
    """frx_traj_max_rates vs Piece::getMaxVelRate / getMaxAccRate of the reference (Sturm-sequence root isolation)"""
    if ob.ref_traj() is None:
        pytest.skip("oracle/_ref/libref_traj.so not built (reference tree absent)")
    rng = np.random.default_rng(3)
    for sid in (2, 6):
        T, Cf = _optimised(sc, ob, sid, 12, 3)
        mv, ma = frx.traj_max_rates(T, Cf)
        pl = ob.piece_layout(Cf)
        for i in range(len(T)):
            out = np.zeros(2)
            ob.ref_traj().ref_piece_max_rates(float(T[i]), np.ascontiguousarray(pl[i].reshape(-1)), out)
            assert abs(mv[i] - out[0]) <= 1e-9 * max(out[0], 1.0) and abs(ma[i] - out[1]) <= 1e-9 * max(out[1], 1.0), (i, mv[i], ma[i], out)
        out3 = np.zeros(3)
        ob.ref_traj().ref_traj_max_rates(len(T), np.ascontiguousarray(T), np.ascontiguousarray(pl.reshape(-1)), out3)
        assert abs(mv.max() - out3[0]) <= 1e-9 * out3[0] and abs(ma.max() - out3[1]) <= 1e-9 * out3[1] and abs(T.sum() - out3[2]) < 1e-12
    # random quintics, incl. constant-velocity and zero pieces
    for trial in range(200):
        c = rng.normal(0, 1, (6, 3)) * rng.choice([0.0, 1.0], (6, 1), p=[0.2, 0.8])
        T1 = np.array([rng.uniform(0.05, 3.0)])
        mv, ma = frx.traj_max_rates(T1, c)
        out = np.zeros(2)
        ob.ref_traj().ref_piece_max_rates(float(T1[0]), np.ascontiguousarray(ob.piece_layout(c)[0].reshape(-1)), out)
        assert abs(mv[0] - out[0]) <= 1e-9 * max(out[0], 1.0) and abs(ma[0] - out[1]) <= 1e-9 * max(out[1], 1.0), (trial, mv, ma, out)


def test_max_rates_bound_dense_sampling(frx, sc, ob):
    """reference-free property: the reported maxima dominate a dense sampling and are attained to 1e-9"""
    T, Cf = _optimised(sc, ob, 4, 8, 2)
    mv, ma = frx.traj_max_rates(T, Cf)
    pc = Cf.reshape(-1, 6, 3)
    k = np.arange(6)
    for i in range(len(T)):
        t = np.linspace(0, T[i], 20001)[:, None]
        v = ((k * t ** np.maximum(k - 1, 0)) @ pc[i]); a = ((k * (k - 1) * t ** np.maximum(k - 2, 0)) @ pc[i])
        sv, sa = np.linalg.norm(v, axis=1).max(), np.linalg.norm(a, axis=1).max()
        assert mv[i] >= sv - 1e-12 and mv[i] <= sv * (1 + 1e-6) + 1e-12
        assert ma[i] >= sa - 1e-12 and ma[i] <= sa * (1 + 1e-6) + 1e-12


def test_wire_format_and_sampling_match_reference_types(frx, sc, ob):
    """frx_traj_to_msg vs Piece::normalizePosCoeffMat, frx_msg_sample vs Trajectory::getPos/getVel/getAcc/getJer of the reference"""
    if ob.ref_traj() is None:
        pytest.skip("oracle/_ref/libref_traj.so not built (reference tree absent)")
    T, Cf = _optimised(sc, ob)
    pl = ob.piece_layout(Cf)
    cx, cy, cz, tm, od = msg = frx.traj_to_msg(T, Cf)
    for i in range(len(T)):
        out = np.zeros(18)
        ob.ref_traj().ref_piece_normalized(float(T[i]), np.ascontiguousarray(pl[i].reshape(-1)), out)
        got = np.stack([cx[6 * i:6 * i + 6], cy[6 * i:6 * i + 6], cz[6 * i:6 * i + 6]])
        assert np.abs(got - out.reshape(3, 6)).max() <= 1e-13 * max(np.abs(out).max(), 1.0)
    rng = np.random.default_rng(1)
    for t in list(rng.uniform(0, T.sum(), 50)) + [0.0, float(T[0]), float(T.sum())]:
        want = [np.zeros(3) for _ in range(4)]
        ob.ref_traj().ref_traj_eval(len(T), np.ascontiguousarray(T), np.ascontiguousarray(pl.reshape(-1)), float(t), *want)
        got = frx.msg_sample(msg, t)
        for g, w, nm in zip(got, want, "pvaj"):
            assert np.abs(g - w).max() <= 1e-9 * max(np.abs(w).max(), 1.0), (t, nm)


def test_corridor_edge_cases(frx):
    """two-point path, empty cloud, blocked-everywhere callback (every segment degenerates to consecutive points), capacity error"""
    bbox = np.array([4.0, 4.0, 2.5])
    cells = frx.corridor_generate(np.array([[0.0, 0.0, 1.0], [0.0, 3.0, 1.0]]), np.zeros((0, 3)), bbox, 3.0)
    assert len(cells) == 1 and cells[0].shape == (6, 8)                       # local box (6 planes) + floor + ceiling
    V = frx.enumerate_vertices(cells[0])
    assert V.shape[1] == 8 and V[2].min() == pytest.approx(0.0) and V[2].max() == pytest.approx(3.0)   # clipped to z in [0, 3]
    path = np.array([[0.0, 0.5 * i, 1.0] for i in range(30)])
    every = frx.corridor_generate(path, np.zeros((0, 3)), bbox, 3.0, blocked=lambda a, b: True)
    free = frx.corridor_generate(path, np.zeros((0, 3)), bbox, 3.0)
    assert len(every) > len(free) >= 2                                       # blocked sight lines shorten the segments
    with pytest.raises(frx.FrxError):
        frx.corridor_generate(path, np.zeros((0, 3)), bbox, 3.0, cap_polys=1)


# ---------------------------------------------------------------------------------------------------------------------
# f1, degenerate inputs: zero-volume and unbounded polytopes (geoutils::findInterior -> SE3GCOPTER::setup == false)
# ---------------------------------------------------------------------------------------------------------------------
def _box(lo, hi):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    cols = []
    for a in range(3):
        n = np.zeros(3); n[a] = 1.0
        cols.append(np.concatenate([n, hi])); cols.append(np.concatenate([-n, lo]))
    return np.array(cols).T                                                # 6 x 6: column = (outer normal; point)


def _enumerate_rc(frx, H):
    import ctypes as C
    nv = C.c_int()
    rc = frx.lib().frx_enumerate_vertices(H.shape[1], np.ascontiguousarray(H.T.reshape(-1)), None, 0, C.byref(nv))
    return rc, nv.value, frx.lib().frx_last_error().decode()


def test_zero_volume_and_unbounded_polytopes_are_rejected_like_the_reference(frx, sc, ob):
    a, b = _box([0, 0, 0], [4, 4, 2]), _box([4, 0, 0], [8, 4, 2])          # two cells that only share the face x = 4
    touching = np.concatenate([a, b], axis=1)
    open_box = a[:, [0, 1, 2, 3, 4, 4]]                                    # the floor plane replaced by a second ceiling: unbounded downwards
    rc, nv, msg = _enumerate_rc(frx, a)
    assert rc == 0 and nv == 8
    rc, nv, msg = _enumerate_rc(frx, touching)
    assert rc == -4 and "no interior" in msg, (rc, nv, msg)                # 4 coplanar vertices used to pass the `< 4` test
    rc, nv, msg = _enumerate_rc(frx, open_box)
    assert rc == -4 and "unbounded" in msg, (rc, nv, msg)
    octant = np.concatenate([a[:, [1, 3, 5]], a[:, [1, 3, 5]]], axis=1)    # x, y, z >= 0 (each plane twice): inscribed radius unbounded
    rc, nv, msg = _enumerate_rc(frx, octant)
    assert rc == -4 and "unbounded" in msg, (rc, nv, msg)
    if ob.ref_gcopter() is None:
        return
    # the reference itself: a single-cell problem sets up for the box and refuses the two degenerate cells
    st = np.zeros((3, 3)); st[:, 0] = (1.0, 1.0, 1.0)
    fin = np.zeros((3, 3)); fin[:, 0] = (3.0, 3.0, 1.0)
    box_v = frx.enumerate_vertices(a)
    ob.Reference(sc.Candidate(st, fin, [a], [box_v]), sc.ZHANGJIAJIE, override_vs=False, qd_intervals=8)
    # (the open box is NOT given to the reference: its Chebyshev LP is bounded there - radius 2 - so findInterior accepts it and
    # enumerateVs then runs quickhull on the polar dual of an unbounded set, which has no defined result)
    for bad in (touching, octant):
        with pytest.raises(RuntimeError):
            ob.Reference(sc.Candidate(st, fin, [bad], [box_v]), sc.ZHANGJIAJIE, override_vs=False, qd_intervals=8)


# ---------------------------------------------------------------------------------------------------------------------
# GPU parity of the "next" rows against the oracle / the compiled reference (the -m gpu tests of f1-f3 were
# self-comparisons)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_handle_built_from_h_polytopes_matches_the_oracle_on_the_device(frx, sc, ob):
    """f1 end to end: the handle gets H-polytopes only (frx_problem_create_from_h enumerates cells and overlaps), the CPU oracle
    gets the generator's V-polytopes: initial guess, objective and gradient agree to the per-evaluation tolerance, and a plan ends
    with the reference's verdict."""
    cands = sc.make_batch(6, 3, 24, 6, obstacles=True)
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=12, enumerate_v=True)
    x0 = prob.initial_guess()
    oracles = [ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=12) for c in cands]
    for o in oracles: o.set_abscissa_mode(False)
    f, g = prob.objective(x0)
    for b, o in enumerate(oracles):
        sl = slice(prob.x_off[b], prob.x_off[b + 1])
        assert np.abs(x0[sl] - o.initial_guess()).max() < 1e-9
        f_ref, g_ref = o.objective(x0[sl])
        assert abs(f[b] - f_ref) <= 1e-9 * abs(f_ref)
        assert np.abs(g[sl] - g_ref).max() <= 1e-9 * max(np.abs(g_ref).max(), abs(f_ref))
    r = prob.optimize(1e-5)
    for b, o in enumerate(oracles):
        assert r["status"][b] == o.optimize(1e-5)["status"]
    prob.close()


@pytest.mark.gpu
def test_device_plan_through_the_wire_format_matches_reference_types(frx, sc, ob):
    """f3 end to end: a trajectory optimised ON THE DEVICE goes through frx_traj_to_msg / frx_msg_sample / frx_traj_max_rates and is
    compared with the reference's own Trajectory / Piece code (trajectory.hpp + root_finder.hpp compiled unmodified,
    oracle/_ref/libref_traj.so) evaluating the same coefficients: normalised coefficients 1e-13, sampled p/v/a/j 1e-9, max rates 1e-6."""
    if ob.ref_traj() is None:
        pytest.skip("oracle/_ref/libref_traj.so not built")
    cands = sc.make_batch(8, 2, 32, 8)
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=8)
    r = prob.optimize(1e-5)
    assert np.all(r["status"] >= 0)
    for b in range(2):
        sl = slice(prob.piece_off[b], prob.piece_off[b + 1])
        T, Cf = r["T"][sl], r["C"][6 * sl.start:6 * sl.stop]
        pl = ob.piece_layout(Cf)
        cx, cy, cz, tm, od = msg = frx.traj_to_msg(T, Cf)
        for i in range(len(T)):
            out = np.zeros(18)
            ob.ref_traj().ref_piece_normalized(float(T[i]), np.ascontiguousarray(pl[i].reshape(-1)), out)
            got = np.stack([cx[6 * i:6 * i + 6], cy[6 * i:6 * i + 6], cz[6 * i:6 * i + 6]])
            assert np.abs(got - out.reshape(3, 6)).max() <= 1e-13 * max(np.abs(out).max(), 1.0)
        rng = np.random.default_rng(b)
        for t in list(rng.uniform(0, T.sum(), 40)) + [0.0, float(T.sum())]:
            want = [np.zeros(3) for _ in range(4)]
            ob.ref_traj().ref_traj_eval(len(T), np.ascontiguousarray(T), np.ascontiguousarray(pl.reshape(-1)), float(t), *want)
            for gv, w in zip(frx.msg_sample(msg, t), want):
                assert np.abs(gv - w).max() <= 1e-9 * max(np.abs(w).max(), 1.0)
        mv, ma = frx.traj_max_rates(T, Cf)
        out3 = np.zeros(3)
        ob.ref_traj().ref_traj_max_rates(len(T), np.ascontiguousarray(T), np.ascontiguousarray(pl.reshape(-1)), out3)
        assert abs(mv.max() - out3[0]) <= 1e-6 * out3[0] and abs(ma.max() - out3[1]) <= 1e-6 * out3[1]
        assert mv.max() <= 1.05 * sc.ZHANGJIAJIE["vel_max"]                  # the optimiser kept the speed limit (soft penalty)
    prob.close()


@pytest.mark.gpu
def test_device_corridor_cells_match_reference_decomp(frx, ob):
    """f2 on the device (frx_dilate_batch, one workgroup per segment) against the reference's LineSegment3D::dilate compiled from its own
    headers (oracle/_ref/libref_decomp.so) and against the library's host form: same half-spaces (as sets: planes tangent at points ON the
    final ellipsoid tie at distance 1), same ellipsoid, for a batch of 24 segments in one 3000-point cloud plus the edge cases of the CPU test
    (vertical segment, empty cloud, tiny clouds)."""
    rng = np.random.default_rng(77)
    base = rng.uniform(-6, 6, (24, 3))
    p1 = base.copy(); p2 = base + rng.normal(0, 1, (24, 3)) * np.array([3.0, 3.0, 0.6])
    p2[3] = p1[3] + np.array([0.0, 0.0, 2.0])                                # vertical segment: degenerate branch of add_local_bbox
    obs = []
    while len(obs) < 3000:
        s = rng.integers(24); u = rng.uniform(-0.3, 1.3)
        q = p1[s] + u * (p2[s] - p1[s]) + rng.normal(0, 2.5, 3)
        d = [np.linalg.norm(q - (p1[k] + np.clip((q - p1[k]) @ (p2[k] - p1[k]) / ((p2[k] - p1[k]) @ (p2[k] - p1[k])), 0, 1) * (p2[k] - p1[k]))) for k in range(24)]
        if min(d) > 0.35:
            obs.append(q)
    obs = np.array(obs)
    bbox = np.array([4.0, 4.0, 2.5])
    cells = frx.dilate_batch(p1, p2, bbox, obs)
    worst_h = worst_c = 0.0
    for s in range(24):
        H, Cm, d = cells[s]
        Hh, Ch, dh = frx.line_segment_dilate(p1[s], p2[s], bbox, obs)          # host form of the library
        assert H.shape == Hh.shape and np.abs(_canon(H) - _canon(Hh)).max() < 1e-9 and np.abs(Cm - Ch).max() < 1e-9
        if ob.ref_decomp() is not None:
            Hr, Cr, dr = ob.ref_line_segment_dilate(p1[s], p2[s], bbox, obs)
            assert H.shape == Hr.shape, (s, H.shape, Hr.shape)
            worst_h = max(worst_h, np.abs(_canon(H) - _canon(Hr)).max()); worst_c = max(worst_c, np.abs(Cm - Cr).max())
            assert np.abs(d - dr).max() < 1e-12
        for q in (p1[s], p2[s], 0.5 * (p1[s] + p2[s])):                        # the segment is inside its cell
            assert np.all(np.einsum("dk,dk->k", H[:3], q[:, None] - H[3:]) <= 1e-9)
    assert worst_h < 1e-9 and worst_c < 1e-9, (worst_h, worst_c)
    print(f"24 cells, {sum(c[0].shape[1] for c in cells)} half-spaces: worst difference to the reference {worst_h:.2e} (planes) {worst_c:.2e} (ellipsoid)")
    for n in (0, 1, 5):                                                      # empty / tiny clouds, zero bounding box
        o = obs[:n]
        (H, Cm, d), = frx.dilate_batch(p1[:1], p2[:1], bbox, o)
        Hh, Ch, dh = frx.line_segment_dilate(p1[0], p2[0], bbox, o)
        assert H.shape == Hh.shape and np.abs(_canon(H) - _canon(Hh)).max() < 1e-9
    (H, Cm, d), = frx.dilate_batch(p1[:1], p2[:1], np.zeros(3), obs[:200])
    Hh, _, _ = frx.line_segment_dilate(p1[0], p2[0], np.zeros(3), obs[:200])
    assert H.shape == Hh.shape and np.abs(_canon(H) - _canon(Hh)).max() < 1e-9
