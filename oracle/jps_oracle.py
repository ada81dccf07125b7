"""TEST INFRASTRUCTURE -- CPU restatement of the reference's path-search front end.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product (fast-racing_amd/csrc/
frx_search.cpp) never does.  Plain Python / IEEE doubles, sized for small maps.

What is restated, with the reference lines it follows:
  * MapUtil<3> queries            src/path_searching/include/jps_collision/map_util.h:320-425
  * JPSPlanner<3>::plan           src/path_searching/src/jps_planner/jps_planner.cpp:333-420
    removeCornerPts / removeLinePts / samplePath                              :54-95 / :98-117 / :118-148
  * GraphSearch (A* mode only)    src/path_searching/src/jps_planner/graph_search.cpp:6-277 with the comparator of
                                  graph_search.h:20-33 and the sift rules of the heap it uses (see boost_shim/)
  * the gate-to-gate splice       src/plan_manage/src/MinCoPlan_CPU.cpp:13-35

Pinning: oracle/_ref/libref_jps.so is the reference's own graph_search.cpp compiled where it lies (recipe: oracle/Makefile, wrapper:
ref_jps_wrap.cpp, heap: boost_shim/); `ref_grid_search` below calls it, and `astar` below is checked against it by
tests/test_front_end.py.  JPSPlanner and MapUtil include ROS, PCL and octomap headers; oracle/ros_shim/ holds empty stand-ins for the
types those headers name, with which jps_planner.cpp and map_util.h compile UNMODIFIED into oracle/_ref/libref_jpsplanner.so
(ref_jpsplanner_wrap.cpp; `RefPlanner` below).  The restatement in this file (Map, plan and the three path filters) is checked against
that library bit for bit by tests/test_front_end.py, and so is the product.
"""
import ctypes as C
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ref = None


def ref_jps():
    """oracle/_ref/libref_jps.so (None when it has not been built)."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref_jps.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        vp = C.c_void_p
        R.ref_grid_search.argtypes = [vp, vp, vp, vp, C.c_double, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]
        R.ref_grid_search.restype = C.c_int
        R.ref_jps_tables.argtypes = [vp] * 6
        _ref = R
    return _ref


def ref_grid_search(cmap, dim, start, goal, eps=1.0, use_jps=False, max_expand=-1, cap=1 << 20):
    """The compiled reference search: (found, path goal-first k x 3, closed-set size, g of the goal)."""
    R = ref_jps()
    cmap = np.ascontiguousarray(cmap, dtype=np.int8)
    dim = np.asarray(dim, np.int32); start = np.asarray(start, np.int32); goal = np.asarray(goal, np.int32)
    path = np.zeros((cap, 3), np.int32); n = C.c_int(); nc = C.c_int(); g = C.c_double()
    ok = R.ref_grid_search(cmap.ctypes.data, dim.ctypes.data, start.ctypes.data, goal.ctypes.data, float(eps), int(use_jps), int(max_expand),
                           path.ctypes.data, cap, C.byref(n), C.byref(nc), C.byref(g))
    return bool(ok), path[:n.value].copy(), nc.value, g.value


def ref_tables():
    R = ref_jps()
    t = [np.zeros(s, dtype=np.int32) for s in ((27, 3, 26), (27, 3, 12), (27, 3, 12), (9, 2, 8), (9, 2, 2), (9, 2, 2))]
    R.ref_jps_tables(*[a.ctypes.data for a in t])
    return t


def c_round(x: float) -> int:
    """std::round: nearest integer, halves away from zero (x - floor(x) is exact in binary floating point)."""
    f = math.floor(x)
    if x >= 0:
        return int(f) + (1 if x - f >= 0.5 else 0)
    return int(f) + (1 if x - f > 0.5 else 0)


class Map:
    """map_util.h: origin_d_, dim_, res_, map_ (0 free, 100 occupied, -1 unknown), x fastest."""

    def __init__(self, origin, dim, res, cells):
        self.origin = [float(v) for v in origin]; self.dim = [int(v) for v in dim]; self.res = float(res)
        self.cells = np.asarray(cells, dtype=np.int8).reshape(-1)

    def outside(self, c):  # :329-334
        return any(c[i] < 0 or c[i] >= self.dim[i] for i in range(3))

    def index(self, c):  # :145-148
        return c[0] + self.dim[0] * c[1] + self.dim[0] * self.dim[1] * c[2]

    def is_free(self, c):  # :336-341
        return (not self.outside(c)) and self.cells[self.index(c)] == 0

    def float_to_int(self, p):  # :382-387
        return [c_round((p[i] - self.origin[i]) / self.res - 0.5) for i in range(3)]

    def int_to_float(self, c):  # :389-392
        return [(float(c[i]) + 0.5) * self.res + self.origin[i] for i in range(3)]

    def ray_trace(self, a, b):  # :395-414
        diff = [b[i] - a[i] for i in range(3)]
        max_diff = int(max(abs(d / self.res) for d in diff) / 0.8)
        if max_diff == 0:
            return []
        s = 1.0 / max_diff
        step = [d * s for d in diff]
        out = []; prev = [-1, -1, -1]
        for n in range(1, max_diff):
            pt = [a[i] + step[i] * n for i in range(3)]
            c = self.float_to_int(pt)
            if self.outside(c):
                break
            if c != prev:
                out.append(c)
            prev = c
        return out

    def is_blocked(self, a, b, val=100):  # :417-425
        return any(self.cells[self.index(c)] >= val for c in self.ray_trace(a, b))

    def mark_cloud(self, pts):  # setObs with expand_size 0, :108-136
        n = 0
        for p in np.asarray(pts, dtype=np.float64).reshape(-1, 3):
            c = self.float_to_int(p)
            if self.outside(c):
                continue
            self.cells[self.index(c)] = 100; n += 1
        return n

    def search_cells(self):  # JPSPlanner::updateMap, jps_planner.cpp:309-323
        return (self.cells > 0).astype(np.int8)


def _norm(a, b):
    return math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2 + (a[2] - b[2]) ** 2)


def remove_corner_pts(m: Map, path):  # jps_planner.cpp:54-95
    if len(path) < 2:
        return list(path)
    out = [path[0]]; prev = path[0]
    cost1 = math.inf if m.is_blocked(path[0], path[1]) else _norm(path[0], path[1])
    for i in range(1, len(path) - 1):
        p1, p2 = path[i], path[i + 1]
        cost2 = math.inf if m.is_blocked(p1, p2) else _norm(p1, p2)
        cost3 = math.inf if m.is_blocked(prev, p2) else _norm(prev, p2)
        if cost3 < cost1 + cost2:
            cost1 = cost3
        else:
            out.append(p1); cost1 = _norm(p1, p2); prev = p1
    out.append(path[-1])
    return out


def remove_line_pts(path):  # jps_planner.cpp:98-117
    if len(path) < 3:
        return list(path)
    out = [path[0]]
    for i in range(1, len(path) - 1):
        p = [(path[i + 1][k] - path[i][k]) - (path[i][k] - path[i - 1][k]) for k in range(3)]
        if abs(p[0]) + abs(p[1]) + abs(p[2]) > 1e-2:
            out.append(path[i])
    out.append(path[-1])
    return out


def sample_path(m: Map, path):  # jps_planner.cpp:118-148
    out = [path[0]]; last = m.float_to_int(path[0])
    for i in range(len(path) - 1):
        p1, p2 = path[i], path[i + 1]
        l = _norm(p2, p1)
        d = 0.0
        while d <= l:
            w1 = (l - d) / l; w2 = d / l
            q = [w1 * p1[k] + w2 * p2[k] for k in range(3)]
            c = m.float_to_int(q)
            if c != last:
                out.append(m.int_to_float(c)); last = c
            d += 0.02
        # the reference's `if (d < l)` after the loop cannot hold
    return out


def astar(cmap, dim, start, goal, eps=1.0, max_expand=-1):
    """GraphSearch::plan with useJps = false (graph_search.cpp:104-277), 3-D: six face neighbours, unit cost.
    Returns (path goal-first as list of cells or [], expansions)."""
    X, Y, Z = dim

    def cid(x, y, z):
        return x + y * X + z * X * Y

    def free(x, y, z):
        return 0 <= x < X and 0 <= y < Y and 0 <= z < Z and cmap[cid(x, y, z)] == 0

    def heur(x, y, z):
        return eps * math.sqrt((x - goal[0]) ** 2 + (y - goal[1]) ** 2 + (z - goal[2]) ** 2)

    nodes = {}  # id -> dict
    heap = []

    def after(a, b):  # graph_search.h:20-33
        f1 = a["g"] + a["h"]; f2 = b["g"] + b["h"]
        if f2 - 0.000001 <= f1 <= f2 + 0.000001:
            return a["g"] < b["g"]
        return f1 > f2

    def swap(i, j):
        heap[i], heap[j] = heap[j], heap[i]
        heap[i]["pos"] = i; heap[j]["pos"] = j

    def up(i):
        while i:
            p = (i - 1) // 2
            if not after(heap[p], heap[i]):
                return
            swap(p, i); i = p

    def down(i):
        while 2 * i + 1 < len(heap):
            c = 2 * i + 1
            if c + 1 < len(heap) and after(heap[c], heap[c + 1]):
                c += 1
            if after(heap[c], heap[i]):
                return
            swap(c, i); i = c

    def push(n):
        heap.append(n); n["pos"] = len(heap) - 1; up(len(heap) - 1)

    def pop():
        top = heap[0]
        swap(0, len(heap) - 1); heap.pop()
        if heap:
            down(0)
        return top

    ns = [(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if abs(x) + abs(y) + abs(z) == 1]
    sid = cid(*start); gid = cid(*goal)
    cur = dict(id=sid, c=tuple(start), g=0.0, h=heur(*start), parent=-1, opened=True, closed=False)
    nodes[sid] = cur; push(cur)
    expanded = 0
    while True:
        expanded += 1
        cur = pop(); cur["closed"] = True
        if cur["id"] == gid:
            break
        for d in ns:
            c = (cur["c"][0] + d[0], cur["c"][1] + d[1], cur["c"][2] + d[2])
            if not free(*c):
                continue
            i = cid(*c)
            ch = nodes.get(i)
            if ch is None:
                ch = dict(id=i, c=c, g=math.inf, h=heur(*c), parent=-1, opened=False, closed=False); nodes[i] = ch
            t = cur["g"] + 1.0
            if t < ch["g"]:
                ch["parent"] = cur["id"]; ch["g"] = t
                if ch["opened"] and not ch["closed"]:
                    up(ch["pos"])
                elif ch["opened"] and ch["closed"]:
                    continue
                else:
                    push(ch); ch["opened"] = True
        if max_expand > 0 and expanded >= max_expand:
            return [], expanded
        if not heap:
            return [], expanded
    path = [cur["c"]]
    while cur["id"] != sid:
        cur = nodes[cur["parent"]]; path.append(cur["c"])
    return path, expanded


def plan(m: Map, start, goal, eps=1.0, use_jps=False, search="auto"):
    """JPSPlanner<3>::plan (jps_planner.cpp:333-420) -> dict(status, raw_path, path, sample_path).
    search = "ref" (compiled reference search), "python" (astar above; A* mode only) or "auto"."""
    res = dict(status=0, raw_path=[], path=[], sample_path=[])
    s = m.float_to_int(start)
    if not m.is_free(s):
        res["status"] = 1; return res
    g = m.float_to_int(goal)
    if not m.is_free(g):
        res["status"] = 2; return res
    cmap = m.search_cells()
    if search == "auto":
        search = "ref" if ref_jps() is not None else "python"
    if search == "ref":
        ok, cells, _, _ = ref_grid_search(cmap, m.dim, s, g, eps, use_jps)
        cells = [list(map(int, c)) for c in cells] if ok else []
    else:
        assert not use_jps, "the Python restatement covers the A* mode (the one plan_manage calls)"
        cells, _ = astar(cmap, m.dim, s, g, eps)
        cells = [list(c) for c in cells]
    if len(cells) < 1:
        res["status"] = -1; return res
    raw = [m.int_to_float(c) for c in cells][::-1]
    p = remove_corner_pts(m, raw)
    p = remove_corner_pts(m, p[::-1])[::-1]
    p = remove_line_pts(p)
    res.update(raw_path=raw, path=p, sample_path=sample_path(m, p))
    return res


def route(m: Map, start, goal, gates, eps=1.0, use_jps=False, search="auto"):
    """MinCoPlan_CPU.cpp:13-35: the sample paths of the legs, each junction kept once.  ([], statuses) when a leg fails."""
    wp = [list(start)] + [list(g) for g in gates] + [list(goal)]
    out = []; st = []
    for a, b in zip(wp[:-1], wp[1:]):
        r = plan(m, a, b, eps, use_jps, search)
        st.append(r["status"])
        if r["status"] != 0:
            continue
        if out:
            out.pop()
        out.extend(r["sample_path"])
    if any(st):
        return [], st
    return out, st


# ---- the reference's own JPSPlanner<3> / MapUtil<3>, compiled where they lie (oracle/_ref/libref_jpsplanner.so) ----
_refp = None


def ref_planner_lib():
    global _refp
    if _refp is None:
        path = os.path.join(HERE, "_ref", "libref_jpsplanner.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        vp = C.c_void_p
        R.ref_jp_create.argtypes = [vp, vp, vp, C.c_double]; R.ref_jp_create.restype = vp
        R.ref_jp_destroy.argtypes = [vp]
        R.ref_jp_mark.argtypes = [vp, vp, C.c_int, vp]
        R.ref_jp_is_blocked.argtypes = [vp, vp, vp]; R.ref_jp_is_blocked.restype = C.c_int
        R.ref_jp_float_to_int.argtypes = [vp, vp, vp]
        R.ref_jp_plan.argtypes = [vp, vp, vp, C.c_double, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int]; R.ref_jp_plan.restype = C.c_int
        _refp = R
    return _refp


class RefPlanner:
    """MapUtil<3>::setMap + JPSPlanner<3> of the reference itself."""

    def __init__(self, origin, dim, res, cells):
        self.R = ref_planner_lib()
        self.dim = [int(d) for d in dim]
        o = np.asarray(origin, np.float64); d = np.asarray(self.dim, np.int32)
        c = np.ascontiguousarray(np.asarray(cells, np.int8).reshape(-1))
        self.h = self.R.ref_jp_create(o.ctypes.data, d.ctypes.data, c.ctypes.data, float(res))

    def close(self):
        if self.h:
            self.R.ref_jp_destroy(self.h); self.h = None

    def mark_cloud(self, pts):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.zeros(self.dim[0] * self.dim[1] * self.dim[2], np.int8)
        self.R.ref_jp_mark(self.h, pts.ctypes.data, len(pts), out.ctypes.data)
        return out

    def is_blocked(self, a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return bool(self.R.ref_jp_is_blocked(self.h, a.ctypes.data, b.ctypes.data))

    def float_to_int(self, p):
        p = np.asarray(p, np.float64); out = np.zeros(3, np.int32)
        self.R.ref_jp_float_to_int(self.h, p.ctypes.data, out.ctypes.data)
        return out.tolist()

    def plan(self, start, goal, eps=1.0, use_jps=False, cap=1 << 16):
        s = np.asarray(start, np.float64); g = np.asarray(goal, np.float64)
        raw = np.zeros((cap, 3)); path = np.zeros((cap, 3)); samp = np.zeros((cap, 3))
        st, nr, np_, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        ok = self.R.ref_jp_plan(self.h, s.ctypes.data, g.ctypes.data, float(eps), int(use_jps), C.byref(st), raw.ctypes.data, C.byref(nr), path.ctypes.data,
                                C.byref(np_), samp.ctypes.data, C.byref(ns), cap)
        return {"ok": bool(ok), "status": st.value, "raw_path": raw[:nr.value].copy(), "path": path[:np_.value].copy(), "sample_path": samp[:ns.value].copy()}
