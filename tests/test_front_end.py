"""the path-search front end (frx_grid_search / frx_jps_plan / frx_route_plan, csrc/frx_search.cpp).

Checkers: oracle/_ref/libref_jps.so = the reference's own graph_search.cpp compiled where it lies (grid search, tables);
oracle/_ref/libref_jpsplanner.so = the reference's own JPSPlanner<3> and MapUtil<3> (jps_planner.cpp, map_util.h) compiled where they lie
against empty ROS / PCL / octomap stand-ins; oracle/jps_oracle.py = restatement of the planner's map queries and path post-processing and
of the A* search, itself checked against the compiled planner below."""
import numpy as np
import pytest

from oracle import jps_oracle as jo

needs_ref = pytest.mark.skipif(jo.ref_jps() is None, reason="oracle/_ref/libref_jps.so not built")

NSZ3 = [(26, 0), (1, 8), (3, 12), (7, 12)]
NSZ2 = [(8, 0), (1, 2), (3, 2)]


def random_grid(rng, three):
    dim = [int(rng.integers(4, 24)), int(rng.integers(4, 24)), int(rng.integers(2, 10))] if three else [int(rng.integers(4, 40)), int(rng.integers(4, 40)), 0]
    n = dim[0] * dim[1] * max(dim[2], 1)
    cmap = (rng.random(n) < rng.choice([0.05, 0.15, 0.3, 0.4])).astype(np.int8) * int(rng.choice([1, 100]))
    free = np.flatnonzero(cmap == 0)
    si, gi = rng.choice(free, 2)
    co = lambda i: [int(i % dim[0]), int((i // dim[0]) % dim[1]), int(i // (dim[0] * dim[1]))]
    return cmap, dim, co(si), co(gi)


@needs_ref
def test_jump_point_tables_equal_the_reference(frx):
    """The rule-generated neighbour tables, entry for entry (the reference leaves the unused tail of each row uninitialised)."""
    ours, ref = frx.jps_tables(), jo.ref_tables()
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                i = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)
                a, b = NSZ3[abs(dx) + abs(dy) + abs(dz)]
                assert np.array_equal(ours[0][i, :, :a], ref[0][i, :, :a])
                assert np.array_equal(ours[1][i, :, :b], ref[1][i, :, :b])
                assert np.array_equal(ours[2][i, :, :b], ref[2][i, :, :b])
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            i = (dx + 1) + 3 * (dy + 1)
            a, b = NSZ2[abs(dx) + abs(dy)]
            assert np.array_equal(ours[3][i, :, :a], ref[3][i, :, :a])
            assert np.array_equal(ours[4][i, :, :b], ref[4][i, :, :b])
            assert np.array_equal(ours[5][i, :, :b], ref[5][i, :, :b])


@needs_ref
def test_grid_search_returns_the_reference_path_cell_for_cell(frx):
    """2-D and 3-D, A* and jump-point search, plain / inflated / deflated heuristic, expansion budget: same verdict, same
    cells in the same order, same cost bits, same number of expansions as the compiled reference."""
    rng = np.random.default_rng(0)
    n = 0
    for trial in range(160):
        cmap, dim, s, g = random_grid(rng, trial % 2 == 0)
        for jps in (False, True):
            for eps, maxe in ((1.0, -1), (1.7, -1), (3.0, -1), (0.5, -1), (1.0, 7)):
                ok, rp, closed, gg = jo.ref_grid_search(cmap, dim, s, g, eps, jps, maxe)
                p, ex, cost = frx.grid_search(cmap, dim, s, g, eps, jps, maxe)
                assert ok == (len(p) > 0)
                assert np.array_equal(p, rp)
                if ok:
                    assert cost == gg and ex == closed
                n += 1
    assert n == 1600


def test_grid_search_equals_the_python_restatement(frx):
    rng = np.random.default_rng(1)
    for trial in range(40):
        cmap, dim, s, g = random_grid(rng, True)
        for eps in (1.0, 2.0):
            want, ex_w = jo.astar(cmap, dim, s, g, eps)
            p, ex, cost = frx.grid_search(cmap, dim, s, g, eps, False)
            assert [tuple(c) for c in p] == [tuple(c) for c in want] and ex == ex_w


def test_search_costs_are_shortest_path_costs(frx):
    """Independent of any restatement: A* (six neighbours, unit steps) returns the breadth-first distance, the jump-point
    search the Dijkstra distance on the 26-neighbour graph, and every path is a chain of free, adjacent cells."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import dijkstra
    rng = np.random.default_rng(2)
    for trial in range(30):
        cmap, dim, s, g = random_grid(rng, True)
        X, Y, Z = dim
        idx = np.arange(X * Y * Z).reshape(Z, Y, X)
        occ = cmap.reshape(Z, Y, X) != 0
        for jps, offs in ((False, [(1, 0, 0), (0, 1, 0), (0, 0, 1)]),
                          (True, [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) > (0, 0, 0)])):
            rows, cols, w = [], [], []
            for dx, dy, dz in offs:
                def sl(d, n):
                    return (slice(max(0, -d), n - max(0, d)), slice(max(0, d), n + min(0, d) if d < 0 else n))
                (ax, bx), (ay, by), (az, bz) = sl(dx, X), sl(dy, Y), sl(dz, Z)
                a = idx[az, ay, ax]; b = idx[bz, by, bx]
                okm = ~(occ[az, ay, ax] | occ[bz, by, bx])
                rows.append(a[okm]); cols.append(b[okm]); w.append(np.full(okm.sum(), np.sqrt(dx * dx + dy * dy + dz * dz)))
            G = coo_matrix((np.concatenate(w), (np.concatenate(rows), np.concatenate(cols))), shape=(X * Y * Z,) * 2)
            si = s[0] + X * s[1] + X * Y * s[2]; gi = g[0] + X * g[1] + X * Y * g[2]
            want = dijkstra(G, directed=False, indices=si)[gi]
            p, ex, cost = frx.grid_search(cmap, dim, s, g, 1.0, jps)
            if np.isinf(want):
                assert len(p) == 0
                continue
            assert cost == pytest.approx(want, rel=1e-12)
            assert tuple(p[0]) == tuple(g) and tuple(p[-1]) == tuple(s)
            for c in p:
                assert not occ[c[2], c[1], c[0]]
            if not jps:
                assert np.all(np.abs(np.diff(p, axis=0)).sum(axis=1) == 1) and len(p) == int(round(want)) + 1
            else:   # jump points: straight runs along one of the 26 directions
                d = np.diff(p, axis=0)
                assert np.all([len(set(np.abs(v[v != 0]))) == 1 for v in d])


def test_grid_search_edge_cases(frx):
    free = np.zeros(5 * 4 * 3, np.int8)
    p, ex, cost = frx.grid_search(free, [5, 4, 3], [1, 1, 1], [1, 1, 1])          # start = goal: one cell, no expansion of neighbours
    assert p.tolist() == [[1, 1, 1]] and cost == 0.0 and ex == 1
    wall = free.copy().reshape(3, 4, 5); wall[:, :, 2] = 1
    p, ex, cost = frx.grid_search(wall, [5, 4, 3], [0, 0, 0], [4, 3, 2])          # sealed off: the whole left half is expanded
    assert len(p) == 0 and ex == 2 * 4 * 3 and cost == -1.0
    for use_jps in (False, True):
        p, _, _ = frx.grid_search(wall, [5, 4, 3], [0, 0, 0], [4, 3, 2], 1.0, use_jps)
        assert len(p) == 0
    with pytest.raises(frx.FrxError) as e:
        frx.grid_search(free, [5, 4, 3], [5, 0, 0], [1, 1, 1])                    # the reference would index out of bounds
    assert e.value.code == -1
    with pytest.raises(frx.FrxError) as e:
        frx.grid_search(free, [5, 4, 3], [0, 0, 0], [4, 3, 2], cap=3)
    assert e.value.code == -5
    p2, _, c2 = frx.grid_search(np.zeros(30, np.int8), [6, 5, 0], [0, 0, 0], [5, 4, 0], 1.0, False)   # 2-D: diagonal steps allowed
    assert c2 == pytest.approx(4 * np.sqrt(2) + 1) and len(p2) == 6


def make_map(frx, rng, dim, res, n_boxes, unknown=False):
    origin = [-dim[0] * res / 2, -1.0, 0.0]
    cells = np.zeros(dim[::-1], np.int8)
    for _ in range(n_boxes):
        w = rng.integers(1, 5, 3)
        lo = [rng.integers(0, max(1, dim[i] - w[i])) for i in range(3)]
        cells[lo[2]:lo[2] + w[2] + 3, lo[1]:lo[1] + w[1], lo[0]:lo[0] + w[0]] = 100
    if unknown:
        cells[rng.random(cells.shape) < 0.02] = -1
    return frx.VoxelMap(origin, dim, res, cells), jo.Map(origin, dim, res, cells.copy())


def free_point(rng, m):
    while True:
        c = [int(rng.integers(0, m.dim[i])) for i in range(3)]
        if m.is_free(c):
            p = m.int_to_float(c)
            return [p[i] + rng.uniform(-0.45, 0.45) * m.res for i in range(3)]


@pytest.mark.parametrize("search", ["python"] + (["ref"] if jo.ref_jps() is not None else []))
def test_planner_paths_equal_the_oracle_bit_for_bit(frx, search):
    """JPSPlanner<3>::plan: raw path, simplified path and sample path against the restatement, on maps with boxes (and
    unknown cells, which the search crosses but start / goal may not sit in)."""
    rng = np.random.default_rng(3)
    n_ok = 0
    for trial in range(24):
        dim = [int(rng.integers(20, 44)), int(rng.integers(30, 70)), int(rng.integers(6, 14))]
        vm, om = make_map(frx, rng, dim, [0.1, 0.25][trial % 2], int(rng.integers(5, 40)), unknown=trial % 3 == 0)
        a, b = free_point(rng, om), free_point(rng, om)
        for use_jps in ((False, True) if search == "ref" else (False,)):
            want = jo.plan(om, a, b, 1.0, use_jps, search)
            got = vm.plan(a, b, 1.0, use_jps)
            assert got["status"] == want["status"]
            for k in ("raw_path", "path", "sample_path"):
                w = np.array(want[k], dtype=np.float64).reshape(-1, 3)
                assert got[k].shape == w.shape and np.array_equal(got[k], w), (trial, k)
            if want["status"] == 0:
                n_ok += 1
                assert np.array_equal(got["sample_path"][0], got["path"][0])
                # samples are cell centres; consecutive ones are distinct cells at most one step apart on every axis
                cs = np.array([om.float_to_int(list(p)) for p in got["sample_path"]])
                assert np.array_equal(got["sample_path"], np.array([om.int_to_float(list(c)) for c in cs]))
                if len(cs) > 1:
                    step = np.abs(np.diff(cs, axis=0))
                    assert step.max() <= 1 and step.sum(axis=1).min() >= 1
    assert n_ok >= 10


needs_ref_planner = pytest.mark.skipif(jo.ref_planner_lib() is None, reason="oracle/_ref/libref_jpsplanner.so not built")


@needs_ref_planner
def test_planner_paths_equal_the_compiled_reference_planner(frx):
    """The product AND the restatement against JPSPlanner<3>::plan itself (jps_planner.cpp:333-420 compiled where it lies): status, raw path,
    simplified path and sample path, coordinate for coordinate, A* and jump-point search, maps with boxes and unknown cells."""
    rng = np.random.default_rng(3)
    n_ok = 0
    for trial in range(24):
        dim = [int(rng.integers(20, 44)), int(rng.integers(30, 70)), int(rng.integers(6, 14))]
        vm, om = make_map(frx, rng, dim, [0.1, 0.25][trial % 2], int(rng.integers(5, 40)), unknown=trial % 3 == 0)
        rp = jo.RefPlanner(om.origin, om.dim, om.res, om.cells)
        a, b = free_point(rng, om), free_point(rng, om)
        for use_jps in (False, True):
            want = rp.plan(a, b, 1.0, use_jps)
            got = vm.plan(a, b, 1.0, use_jps)
            mine = jo.plan(om, a, b, 1.0, use_jps, "ref" if jo.ref_jps() is not None else "python") if (jo.ref_jps() is not None or not use_jps) else None
            assert got["status"] == want["status"]
            for k in ("raw_path", "path", "sample_path"):
                assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), (trial, use_jps, k)
                if mine is not None:
                    assert np.array_equal(np.array(mine[k], dtype=np.float64).reshape(-1, 3), want[k]), (trial, use_jps, k, "restatement")
            n_ok += want["status"] == 0
        # start / goal verdicts: occupied start, goal outside
        occ = np.argwhere(np.asarray(om.cells).reshape(dim[::-1]) == 100)
        if len(occ):
            z, y, x = occ[0]
            bad = om.int_to_float([int(x), int(y), int(z)])
            assert vm.plan(bad, b)["status"] == rp.plan(bad, b)["status"] == 1
            assert vm.plan(a, bad)["status"] == rp.plan(a, bad)["status"] == 2
        rp.close()
    assert n_ok >= 20


@needs_ref_planner
def test_map_queries_equal_the_compiled_reference_map(frx):
    """MapUtil<3>: setObs (cloud marking, points on cell faces included), floatToInt and isBlocked / rayTrace (map_util.h:108-136, 382-425) of the
    reference itself against the product and the restatement."""
    rng = np.random.default_rng(5)
    vm = frx.VoxelMap.from_params(6.0, 9.0, 1.5, 0.1)
    om = jo.Map(vm.origin, vm.dim, vm.res, np.zeros(vm.cells.size, np.int8))
    rp = jo.RefPlanner(vm.origin, vm.dim, vm.res, np.zeros(vm.cells.size, np.int8))
    pts = np.column_stack([rng.uniform(-3.4, 3.4, 3000), rng.uniform(-10.5, -0.5, 3000), rng.uniform(-0.2, 1.7, 3000)])
    pts[:50, 0] = np.round(pts[:50, 0], 1)                                        # points exactly on cell faces
    vm.mark_cloud(pts); om.mark_cloud(pts)
    ref_cells = rp.mark_cloud(pts)
    assert np.array_equal(np.asarray(vm.cells).reshape(-1), ref_cells) and np.array_equal(np.asarray(om.cells).reshape(-1), ref_cells)
    n_blocked = 0
    for _ in range(600):
        a = [rng.uniform(-3.2, 3.2), rng.uniform(-10.2, -0.8), rng.uniform(-0.1, 1.6)]
        b = [a[0] + rng.normal(0, 1.0), a[1] + rng.normal(0, 1.5), a[2] + rng.normal(0, 0.3)] if rng.random() < 0.9 else list(a)
        want = rp.is_blocked(a, b)
        assert vm.is_blocked(a, b) == want and om.is_blocked(a, b) == want
        assert om.float_to_int(a) == rp.float_to_int(a)
        n_blocked += want
    assert 30 < n_blocked < 570
    rp.close()


def test_planner_status_codes(frx):
    rng = np.random.default_rng(4)
    dim = [20, 30, 8]
    cells = np.zeros(dim[::-1], np.int8)
    cells[:, 15, :] = 100                                   # a full wall across y
    cells[3, 3, 3] = -1                                     # an unknown cell
    vm = frx.VoxelMap([-1.0, -1.0, 0.0], dim, 0.1, cells); om = jo.Map([-1.0, -1.0, 0.0], dim, 0.1, cells)
    inside = lambda c: om.int_to_float(c)
    assert vm.plan(inside([2, 15, 2]), inside([2, 2, 2]))["status"] == 1         # start occupied
    assert vm.plan(inside([3, 3, 3]), inside([2, 2, 2]))["status"] == 1          # start unknown: not free either
    assert vm.plan([5.0, 0.0, 0.3], inside([2, 2, 2]))["status"] == 1            # start outside the map
    assert vm.plan(inside([2, 2, 2]), inside([2, 15, 2]))["status"] == 2         # goal occupied
    r = vm.plan(inside([2, 2, 2]), inside([2, 25, 2]))
    assert r["status"] == -1 and len(r["raw_path"]) == 0 and len(r["sample_path"]) == 0   # wall: no path
    assert jo.plan(om, inside([2, 2, 2]), inside([2, 25, 2]), search="python")["status"] == -1
    r = vm.plan(inside([2, 2, 2]), inside([2, 2, 2]))                             # same cell: a single point everywhere
    assert r["status"] == 0 and len(r["raw_path"]) == 1 and len(r["path"]) == 1 and len(r["sample_path"]) == 1
    import ctypes as C
    st = C.c_int()
    assert frx.lib().frx_jps_plan(None, None, None, 1.0, 0, 0, None, None, None, None, None, None, C.byref(st), None) == -1
    bad = frx.VoxelMap([0, 0, 0], [2, 2, 2], 0.1); bad._s.res = 0.0
    p = np.zeros(3)
    assert frx.lib().frx_jps_plan(C.byref(bad._s), p.ctypes.data, p.ctypes.data, 1.0, 0, 0, None, None, None, None, None, None, C.byref(st), None) == -1
    with pytest.raises(frx.FrxError) as e:                                       # outputs too small: counts are still reported
        vm.plan(inside([2, 2, 2]), inside([17, 10, 6]), cap=2)
    assert e.value.code == -5


def test_map_from_cloud_and_sight_lines(frx):
    rng = np.random.default_rng(5)
    vm = frx.VoxelMap.from_params(6.0, 9.0, 1.5, 0.1)
    assert vm.dim.tolist() == [60, 90, 15] and vm.origin.tolist() == [-3.0, -10.0, 0.0]
    om = jo.Map(vm.origin, vm.dim, vm.res, np.zeros(vm.cells.size, np.int8))
    pts = np.column_stack([rng.uniform(-3.4, 3.4, 3000), rng.uniform(-10.5, -0.5, 3000), rng.uniform(-0.2, 1.7, 3000)])
    pts[:50, 0] = np.round(pts[:50, 0], 1)                                        # points exactly on cell faces
    assert vm.mark_cloud(pts) == om.mark_cloud(pts)
    assert np.array_equal(vm.cells, om.cells) and 0 < (vm.cells == 100).sum() < 3000
    n_blocked = 0
    for _ in range(400):
        a = [rng.uniform(-3.2, 3.2), rng.uniform(-10.2, -0.8), rng.uniform(-0.1, 1.6)]
        b = [a[0] + rng.normal(0, 1.0), a[1] + rng.normal(0, 1.5), a[2] + rng.normal(0, 0.3)] if rng.random() < 0.9 else list(a)
        assert vm.is_blocked(a, b) == om.is_blocked(a, b)
        n_blocked += om.is_blocked(a, b)
    assert 20 < n_blocked < 380


def test_route_through_gates(frx):
    """MinCoPlan_CPU.cpp:13-35: the legs' sample paths spliced; the same on one thread and on many; a failed leg is reported."""
    rng = np.random.default_rng(6)
    dim = [40, 120, 10]
    vm, om = make_map(frx, rng, dim, 0.1, 30)
    pts = [free_point(rng, om) for _ in range(5)]
    start, goal, gates = pts[0], pts[-1], pts[1:-1]
    want, st_w = jo.route(om, start, goal, gates, search="python")
    got1, st1, ex1 = vm.route(start, goal, gates, threads=1)
    got4, st4, ex4 = vm.route(start, goal, gates, threads=4)
    assert st1.tolist() == st_w == st4.tolist() and np.array_equal(ex1, ex4)
    assert np.array_equal(got1, np.array(want).reshape(-1, 3)) and np.array_equal(got1, got4)
    assert len(got1) > 0 and np.array_equal(got1[0], np.array(om.int_to_float(om.float_to_int(start))))   # cell centres, not the query points
    # no gates = one leg = the planner's sample path
    one, st, _ = vm.route(start, goal)
    assert np.array_equal(one, vm.plan(start, goal)["sample_path"]) and st.tolist() == [0]
    # a gate inside an obstacle: leg 1 cannot end there (status 2), leg 2 cannot start there (status 1)
    occ = np.argwhere(om.cells.reshape(dim[::-1]) == 100)[0][::-1]
    bad_gate = om.int_to_float([int(v) for v in occ])
    path, st, _ = vm.route(start, goal, [bad_gate])
    assert len(path) == 0 and st.tolist() == [2, 1]
    assert jo.route(om, start, goal, [bad_gate], search="python") == ([], [2, 1])


def test_corridor_along_a_planned_route_uses_the_grid_sight_lines(frx):
    """The hand-off to the next stage (MinCoPlan_CPU.cpp:37-105): the route is the polyline of frx_corridor_generate and
    frx_map_is_blocked its `blocked` test; the same corridor comes out as with the oracle's isBlocked as a Python callback."""
    rng = np.random.default_rng(7)
    dim = [50, 160, 12]
    vm, om = make_map(frx, rng, dim, 0.1, 40)
    start, goal = free_point(rng, om), free_point(rng, om)
    path, st, _ = vm.route(start, goal)
    assert st.tolist() == [0] and len(path) > 10
    occ = np.argwhere(om.cells.reshape(dim[::-1]) == 100)[:, ::-1]
    cloud = np.array([om.int_to_float([int(v) for v in c]) for c in occ])
    bbox = np.array([1.0, 1.0, 0.5])
    height = dim[2] * 0.1
    native = frx.corridor_generate(path, cloud, bbox, height, blocked=vm)
    via_py = frx.corridor_generate(path, cloud, bbox, height, blocked=lambda a, b: om.is_blocked(list(a), list(b)))
    assert len(native) == len(via_py) >= 1
    for A, B in zip(native, via_py):
        assert np.array_equal(A, B)
    # every route point lies inside at least one cell of the corridor (outer normals n, points p: n.(x - p) <= 0)
    for x in path[:: max(1, len(path) // 40)]:
        assert any(np.all(np.einsum("ik,ik->k", H[:3], x[:, None] - H[3:]) <= 1e-9) for H in native)


@pytest.mark.gpu
def test_voxel_map_to_trajectory_on_the_device(frx, sc):
    """The whole chain of MavGlobalPlanner::plan through the C ABI: point cloud -> voxel map -> route through gates (f4) ->
    corridor with the grid's sight-line test (f2) -> vertices (f1) -> device optimiser -> message (f3).  The flown trajectory
    starts and ends where asked, stays inside the corridor (soft-constraint margin) and never enters an occupied cell."""
    rng = np.random.default_rng(11)
    vm = frx.VoxelMap.from_params(16.0, 46.0, 2.8, 0.1)                          # origin (-8, -10, 0)
    pillars = []
    while len(pillars) < 45:
        c = np.array([rng.uniform(-7, 7), rng.uniform(-6, 32)])
        if all(np.linalg.norm(c - q) > 2.2 for q in pillars):
            pillars.append(c)
    cloud = np.array([[c[0] + dx, c[1] + dy, z] for c in pillars for dx in np.arange(-0.3, 0.31, 0.1) for dy in np.arange(-0.3, 0.31, 0.1)
                      for z in np.arange(0.05, 2.8, 0.1)])
    assert vm.dim.tolist() == [160, 460, 27]                                     # int(2.8 / 0.1) = 27, as in the reference
    assert vm.mark_cloud(cloud) == (cloud[:, 2] < 2.7).sum()                      # the top layer of points is above the map
    om = jo.Map(vm.origin, vm.dim, vm.res, vm.cells)

    def clear_point(y):
        while True:
            p = np.array([rng.uniform(-5, 5), y + rng.uniform(-1, 1), rng.uniform(1.0, 2.0)])
            if all(np.linalg.norm(p[:2] - q) > 1.2 for q in pillars):
                return p
    start, goal = clear_point(-8.0), clear_point(34.0)
    gates = [clear_point(y) for y in (3.0, 13.0, 24.0)]
    route, st, expanded = vm.route(start, goal, gates)
    assert st.tolist() == [0, 0, 0, 0] and len(route) > 300
    occ_cloud = np.array([om.int_to_float([int(v) for v in c]) for c in np.argwhere(vm.cells.reshape(vm.dim[::-1]) == 100)[:, ::-1]])
    polys = frx.corridor_generate(route, occ_cloud, np.array([4.0, 4.0, 2.5]), 2.7, blocked=vm)
    assert 4 <= len(polys) <= 60
    ini = np.zeros((3, 3)); ini[:, 0] = route[0]
    fin = np.zeros((3, 3)); fin[:, 0] = route[-1]
    cand = sc.Candidate(ini_state=ini, fin_state=fin, h_polys=polys, v_polys=[], gates=np.zeros((0, 3)))
    prob = frx.Problem([cand], sc.ZHANGJIAJIE, qd_intervals=8, enumerate_v=True)
    r = prob.optimize(1e-5)
    assert r["status"][0] >= 0, r["status"]
    T = r["T"][:prob.P]; Cf = r["C"][:6 * prob.P]
    msg = frx.traj_to_msg(T, Cf)
    p0, v0, _, _ = frx.msg_sample(msg, 0.0)
    pe, ve, _, _ = frx.msg_sample(msg, float(T.sum()))
    assert np.abs(p0 - route[0]).max() < 1e-6 and np.abs(pe - route[-1]).max() < 1e-6 and np.abs(v0).max() < 1e-6 and np.abs(ve).max() < 1e-6
    hits = 0
    for t in np.linspace(0, float(T.sum()), 600):
        p, _, _, _ = frx.msg_sample(msg, float(t))
        assert any(np.all(np.einsum("dk,dk->k", H[:3], p[:, None] - H[3:]) <= 0.3) for H in polys)
        c = om.float_to_int(list(p))
        hits += (not om.outside(c)) and om.cells[om.index(c)] == 100
    assert hits == 0
    print(f"route {len(route)} cells ({int(expanded.sum())} expansions), {len(polys)} corridor cells, {prob.P} pieces, "
          f"{int(r['iters'][0])} iterations, flight time {T.sum():.2f} s")
    prob.close()
