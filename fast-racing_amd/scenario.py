"""Deterministic synthetic gate / corridor scenarios.

The reference gets its gates from the AirSim binary at run time (se3_node_cpu.cpp:41-62) and
its corridor from JPS + ellipsoid decomposition over a voxel map (MinCoPlan_CPU.cpp:13-105);
neither input exists in the repository.  This module generates stand-ins with the Zhangjiajie
parameters and map extents (zhangjiajie.launch:32-34, zhangjiajie_params.yaml):

  gates → polyline start→gates→goal → N equal-arc-length segments → per segment the local
  bounding box of decomp_util (line_segment.h:47-85, PolyhedronBox [4,4,2.5]) intersected with
  the height planes z∈[0,MapHeight] (MinCoPlan_CPU.cpp:87-90) → H-polytopes (6×K, column =
  (outer normal, point)); V-polytopes of every box and every consecutive overlap by
  deterministic brute-force triple-plane enumeration (stand-in for extractVs, CPU.hpp:1031-1074).

Everything is seeded with SplitMix64 (seed = 20260925 + scenario_id); no global RNG state.
Used identically by the CPU oracle tests, the GPU parity tests and bench.py.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field

import numpy as np

BASE_SEED = 20260925
MASK64 = (1 << 64) - 1

# stock Zhangjiajie parameters (src/plan_manage/misc/zhangjiajie_params.yaml)
ZHANGJIAJIE = dict(
    rho=1000.0, total_t=0.0, grid_res=float("inf"), qd_intervals=48, c2_diffeo=1,
    horiz_half_len=0.5, vert_half_len=0.15, safe_margin=0.08,
    vel_max=14.0, thr_acc_min=5.0, thr_acc_max=12.0, body_rate_max=3.8, grav_acc=9.81,
    penalty_pvtb=(1.0e7, 1.0e4, 1.0e4, 1.0e4), opt_rel_tol=1.0e-6,
    map_height=3.0, polyhedron_box=(4.0, 4.0, 2.5),
)


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self, lo: float = 0.0, hi: float = 1.0) -> float:
        return lo + (hi - lo) * ((self.next_u64() >> 11) * (1.0 / (1 << 53)))

    def normal(self) -> float:
        u1 = max(self.uniform(), 1e-300)
        u2 = self.uniform()
        return float(np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2))

    def randint(self, n: int) -> int:
        return int(self.next_u64() % n)


@dataclass
class Candidate:
    """One trajectory-optimisation problem = the arguments of SE3GCOPTER::setup (CPU.hpp:1076)."""
    ini_state: np.ndarray              # 3x3, columns (p, v, a)
    fin_state: np.ndarray
    h_polys: list                      # coarseN arrays 6 x K_i (column = (n; p)), normals NOT necessarily unit
    v_polys: list                      # 2*coarseN-1 arrays 3 x nv (vertices; [box0, overlap01, box1, ...])
    gates: np.ndarray = field(default=None)

    @property
    def coarse_n(self) -> int:
        return len(self.h_polys)

    def packed(self):
        """CSR packing used by both C ABIs: (h_off, h_rec[6*sumK], v_off, v_rec[3*sumNv])."""
        h_off = np.zeros(len(self.h_polys) + 1, dtype=np.int32)
        for i, h in enumerate(self.h_polys):
            h_off[i + 1] = h_off[i] + h.shape[1]
        h_rec = np.concatenate([h.T.reshape(-1) for h in self.h_polys]).astype(np.float64)
        v_off = np.zeros(len(self.v_polys) + 1, dtype=np.int32)
        for i, v in enumerate(self.v_polys):
            v_off[i + 1] = v_off[i] + v.shape[1]
        v_rec = np.concatenate([v.T.reshape(-1) for v in self.v_polys]).astype(np.float64)
        return h_off, h_rec, v_off, v_rec


def make_gates(rng: SplitMix64, n_gates: int) -> np.ndarray:
    """Gate centres inside the Zhangjiajie volume x∈[−35,35], y∈[−10,390], z∈[0,2.8]."""
    g = np.zeros((n_gates, 3))
    g[0] = (0.0, 10.0, 1.5)
    for k in range(1, n_gates):
        dy = rng.uniform(10.0, 15.0)
        x = min(30.0, max(-30.0, g[k - 1, 0] + rng.uniform(-6.0, 6.0)))
        z = rng.uniform(1.0, 2.3)
        g[k] = (x, g[k - 1, 1] + dy, z)
    return g


def resample_polyline(pts: np.ndarray, n_seg: int) -> np.ndarray:
    """n_seg+1 points at equal arc length along the polyline."""
    seg = np.linalg.norm(np.diff(pts, axis=0), axis=1)
    s = np.concatenate([[0.0], np.cumsum(seg)])
    out = np.zeros((n_seg + 1, 3))
    for m in range(n_seg + 1):
        t = s[-1] * m / n_seg
        j = min(int(np.searchsorted(s, t, side="right")) - 1, len(seg) - 1)
        j = max(j, 0)
        w = 0.0 if seg[j] == 0 else (t - s[j]) / seg[j]
        out[m] = pts[j] + w * (pts[j + 1] - pts[j])
    out[0], out[-1] = pts[0], pts[-1]
    return out


def segment_box(p1: np.ndarray, p2: np.ndarray, bbox, map_height: float) -> np.ndarray:
    """H-polytope (6 x 8) of one corridor cell: add_local_bbox (line_segment.h:47-85) plus the two
    height planes added by MavGlobalPlanner::plan (MinCoPlan_CPU.cpp:87-90)."""
    d = (p2 - p1) / np.linalg.norm(p2 - p1)
    dh = np.array([d[1], -d[0], 0.0])
    if np.linalg.norm(dh) == 0:
        dh = np.array([-1.0, 0.0, 0.0])
    dh = dh / np.linalg.norm(dh)
    dv = np.cross(d, dh)
    cols = [
        (dh, p1 + dh * bbox[1]), (-dh, p1 - dh * bbox[1]),
        (d, p2 + d * bbox[0]), (-d, p1 - d * bbox[0]),
        (dv, p1 + dv * bbox[2]), (-dv, p1 - dv * bbox[2]),
        (np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, map_height])),
        (np.array([0.0, 0.0, -1.0]), np.array([0.0, 0.0, 0.0])),
    ]
    return np.array([np.concatenate([n, p]) for n, p in cols]).T


def obstacle_planes(rng: SplitMix64, p1, p2, count: int) -> np.ndarray:
    """`count` extra half-spaces tangent to a random ellipsoid around the segment (varies K_i)."""
    mid = 0.5 * (p1 + p2)
    half = 0.5 * np.linalg.norm(p2 - p1)
    cols = []
    for _ in range(count):
        n = np.array([rng.normal(), rng.normal(), 0.35 * rng.normal()])
        n /= np.linalg.norm(n)
        axes = np.array([half + rng.uniform(1.2, 2.5), rng.uniform(1.2, 2.5), rng.uniform(0.6, 1.0)])
        d = (p2 - p1) / np.linalg.norm(p2 - p1)
        dh = np.array([d[1], -d[0], 0.0]); dh /= max(np.linalg.norm(dh), 1e-12)
        dv = np.cross(d, dh)
        R = np.stack([d, dh, dv], axis=1)
        support = np.linalg.norm(axes * (R.T @ n))       # support function of the ellipsoid along n
        cols.append(np.concatenate([n, mid + n * support]))
    return np.array(cols).T.reshape(6, -1)


_TRIPLE_CACHE: dict = {}


def _triples(K: int) -> np.ndarray:
    if K not in _TRIPLE_CACHE:
        _TRIPLE_CACHE[K] = np.array(list(itertools.combinations(range(K), 3)), dtype=np.int64)
    return _TRIPLE_CACHE[K]


def enumerate_vertices(hpoly: np.ndarray, tol: float = 1e-9, quant: float = 1e-7) -> np.ndarray:
    """Vertices (3 x nv) of {x : n_k·(x − p_k) ≤ 0} by solving every plane triple (Cramer),
    keeping the feasible solutions, de-duplicating on a `quant` grid and sorting
    lexicographically by grid key (so v0 and the vertex order are deterministic)."""
    n = hpoly[:3].T / np.linalg.norm(hpoly[:3], axis=0)[:, None]      # K x 3 unit normals
    dd = np.einsum("kd,kd->k", n, hpoly[3:].T)                         # K offsets: n·x ≤ d
    tri = _triples(n.shape[0])
    A = n[tri]                                                         # M x 3 x 3
    b = dd[tri]                                                        # M x 3
    det = np.linalg.det(A)
    ok = np.abs(det) > 1e-10
    A, b = A[ok], b[ok]
    x = np.linalg.solve(A, b[..., None])[..., 0]                       # M' x 3
    feas = np.all(x @ n.T <= dd[None, :] + tol, axis=1)
    x = x[feas]
    if x.shape[0] == 0:
        return np.zeros((3, 0))
    key = np.round(x / quant).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    first = np.sort(first)
    x, key = x[first], key[first]
    order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))             # on the grid keys: immune to last-bit noise in ties
    return x[order].T.copy()


def make_candidate(scenario_id: int, n_pieces: int, n_gates: int, perturb_id: int = 0,
                   obstacles: bool = False, params: dict = ZHANGJIAJIE) -> Candidate:
    """Candidate `perturb_id` of scenario `scenario_id`: 0 = nominal gates, >0 = i.i.d. N(0, 0.5 m)
    perturbation of every gate centre ("random gate perturbations", BASELINE.json configs[3])."""
    rng = SplitMix64(BASE_SEED + scenario_id)
    start = np.array([0.0, 0.0, 1.0])                    # se3_node_cpu.cpp:18
    if n_gates == 0:                                     # degenerate test case: one "gate" 4 m per piece ahead, no goal leg
        gates = np.array([start + np.array([0.5 * rng.uniform(-1, 1), 4.0 * n_pieces, 0.3 * rng.uniform(-1, 1)])])
    else:
        gates = make_gates(rng, n_gates)
    if perturb_id > 0:
        prng = SplitMix64((BASE_SEED + scenario_id) * 1000003 + perturb_id)
        for k in range(len(gates)):
            gates[k] += 0.5 * np.array([prng.normal(), prng.normal(), 0.3 * prng.normal()])
            gates[k, 2] = min(2.4, max(0.8, gates[k, 2]))
    goal = gates[-1] + np.array([0.0, 15.0, 0.0]) if n_gates > 0 else gates[-1]
    poly = np.vstack([start, gates, goal]) if n_gates > 0 else np.vstack([start, goal])
    knots = resample_polyline(poly, n_pieces)
    seg_len = np.linalg.norm(np.diff(knots, axis=0), axis=1)
    assert seg_len.max() <= 6.0, "segments must stay short enough for consecutive boxes to overlap"
    orng = SplitMix64((BASE_SEED + scenario_id) * 7919 + 17 * perturb_id + 1)
    bbox, mh = params["polyhedron_box"], params["map_height"]
    h_polys = []
    for i in range(n_pieces):
        h = segment_box(knots[i], knots[i + 1], bbox, mh)
        if obstacles:
            r = orng.randint(7)
            if r:
                h = np.concatenate([h, obstacle_planes(orng, knots[i], knots[i + 1], r)], axis=1)
        h_polys.append(h)
    v_polys = []
    for i in range(n_pieces):
        v_polys.append(enumerate_vertices(h_polys[i]))
        if i + 1 < n_pieces:
            v_polys.append(enumerate_vertices(np.concatenate([h_polys[i], h_polys[i + 1]], axis=1)))
    for v in v_polys:
        assert v.shape[1] >= 4, "corridor cell or overlap has empty interior"
    ini = np.zeros((3, 3)); ini[:, 0] = start            # PVA = (p, 0, 0), se3_node_cpu.cpp:98-99
    fin = np.zeros((3, 3)); fin[:, 0] = goal
    return Candidate(ini, fin, h_polys, v_polys, gates)


def make_batch(scenario_id: int, B: int, n_pieces: int, n_gates: int, obstacles: bool = False,
               independent: bool = False, params: dict = ZHANGJIAJIE) -> list:
    """B candidates.  independent=False: candidate 0 nominal + B−1 gate perturbations of ONE
    scenario (configs 2-4); independent=True: B different scenario ids (Monte-Carlo, config 5)."""
    if independent:
        return [make_candidate(scenario_id + b, n_pieces, n_gates, 0, obstacles, params) for b in range(B)]
    return [make_candidate(scenario_id, n_pieces, n_gates, b, obstacles, params) for b in range(B)]


# the BASELINE.json configurations: name -> (B, N pieces, gates, kappa)
CONFIGS = {
    "plumbing": (1, 64, 16, 48),
    "synthetic8": (8, 32, 8, 8),
    "headline": (32, 64, 16, 16),
    "perturbed256": (256, 64, 16, 16),
    "montecarlo4096": (4096, 64, 16, 16),
}
