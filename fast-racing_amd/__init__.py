"""fast-racing_amd — MI355X-native back-end for Fast-Racing's SE(3) MINCO trajectory optimiser.

Python here is plumbing only: a ctypes mirror of the C ABI in include/frx.h (the drop-in
boundary) plus the synthetic scenario generator.  All arithmetic of the hot path runs in
libfrx.so (HIP kernels for gfx950 + the host L-BFGS driver); there is no Python or CPU fallback,
and importing this package fails loudly when the library has not been built.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import scenario  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfrx.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class FrxConfig(C.Structure):
    """struct frx_config (include/frx.h) = scalar arguments of SE3GCOPTER::setup (CPU.hpp:1076-1092)."""
    _fields_ = [
        ("rho", C.c_double), ("total_t", C.c_double), ("grid_res", C.c_double),
        ("qd_intervals", C.c_int), ("c2_diffeo", C.c_int),
        ("horiz_half_len", C.c_double), ("vert_half_len", C.c_double), ("safe_margin", C.c_double),
        ("vel_max", C.c_double), ("thr_acc_min", C.c_double), ("thr_acc_max", C.c_double),
        ("body_rate_max", C.c_double), ("grav_acc", C.c_double),
        ("penalty_pvtb", C.c_double * 4),
    ]

    @classmethod
    def from_params(cls, params: dict, **override):
        p = dict(params); p.update(override)
        c = cls()
        for name, _ in cls._fields_:
            if name == "penalty_pvtb":
                c.penalty_pvtb = (C.c_double * 4)(*p["penalty_pvtb"])
            else:
                setattr(c, name, p[name])
        return c


class LbfgsParams(C.Structure):
    """struct frx_lbfgs_params = lbfgs::lbfgs_parameter_t (lbfgs.hpp:18-140)."""
    _fields_ = [
        ("mem_size", C.c_int), ("g_epsilon", C.c_double), ("past", C.c_int), ("delta", C.c_double),
        ("max_iterations", C.c_int), ("max_linesearch", C.c_int), ("min_step", C.c_double), ("max_step", C.c_double),
        ("f_dec_coeff", C.c_double), ("s_curv_coeff", C.c_double), ("xtol", C.c_double),
    ]


BATCH_EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                            C.POINTER(C.c_double))

# every symbol include/frx.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "frx_version", "frx_last_error", "frx_device_count", "frx_lbfgs_default_params", "frx_lbfgs_gcopter_params",
    "frx_problem_create", "frx_problem_destroy", "frx_problem_set_solver", "frx_problem_set_lbfgs_mode", "frx_problem_totals", "frx_problem_layout", "frx_initial_guess",
    "frx_objective_eval", "frx_objective_eval_device", "frx_penalty_eval", "frx_penalty_eval_device", "frx_forward",
    "frx_optimize", "frx_optimize_stats", "frx_lbfgs_minimize_batch",
    "frx_problem_create_from_h", "frx_enumerate_vertices", "frx_traj_to_msg", "frx_msg_sample", "frx_line_segment_dilate", "frx_corridor_generate", "frx_traj_max_rates", "frx_objective_eval_async", "frx_wait",
    "frx_problem_set_resident", "frx_optimize_path", "frx_penalty_problem_create", "frx_eval_status",
    "frx_dilate_batch", "frx_multi_create", "frx_multi_destroy", "frx_multi_info", "frx_multi_layout", "frx_multi_initial_guess", "frx_multi_optimize", "frx_multi_last_exchange",
    "frx_map_mark_cloud", "frx_map_is_blocked", "frx_grid_search", "frx_jps_plan", "frx_route_plan",
]
# diagnostics, include/frx_debug.h: not part of the drop-in boundary
DEBUG_SYMBOLS = [
    "frx_debug_trace", "frx_resident_profile", "frx_debug_direction_log", "frx_debug_direction_log_read", "frx_debug_set_resident_retry",
    "frx_debug_resident_counts", "frx_debug_resident_clusters", "frx_debug_resident_predictions", "frx_eval_stage_times", "frx_profile_phases", "frx_dv_selftest", "frx_jps_tables", "frx_debug_host_cpu_share", "frx_debug_taken_over", "frx_debug_compact_from_history",
    "frx_debug_set_eval_fused", "frx_debug_eval_fused", "frx_debug_set_eval_solo", "frx_debug_eval_solo", "frx_debug_penalty_kernel", "frx_debug_mailbox_numa", "frx_eval_launch_time", "frx_debug_profile_eval_cluster", "frx_debug_set_takeover_at", "frx_debug_shader_clock",
]

_lib = None


def lib():
    """Load libfrx.so (built in-tree by __graft_entry__.build() / make -C fast-racing_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                               "There is no fallback implementation.")
        L = C.CDLL(LIB_PATH)
        L.frx_version.restype = C.c_int
        L.frx_last_error.restype = C.c_char_p
        L.frx_device_count.restype = C.c_int
        L.frx_lbfgs_default_params.argtypes = [C.POINTER(LbfgsParams)]
        L.frx_lbfgs_gcopter_params.argtypes = [C.POINTER(LbfgsParams), C.c_double]
        L.frx_problem_create.argtypes = [C.POINTER(FrxConfig), C.c_int, C.c_int, _ip, _dp, _dp, _ip, _dp, _ip, _dp, C.POINTER(C.c_void_p)]
        L.frx_problem_create_from_h.argtypes = [C.POINTER(FrxConfig), C.c_int, C.c_int, _ip, _dp, _dp, _ip, _dp, C.POINTER(C.c_void_p)]
        L.frx_enumerate_vertices.argtypes = [C.c_int, _dp, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        _up = np.ctypeslib.ndpointer(dtype=np.uint32, flags='C_CONTIGUOUS')
        L.frx_traj_to_msg.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _up]
        L.frx_msg_sample.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _up, C.c_double, _dp, _dp, _dp, _dp]
        L.frx_dv_selftest.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.frx_line_segment_dilate.argtypes = [_dp, _dp, _dp, C.c_int, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
        L.frx_corridor_generate.argtypes = [C.c_int, _dp, C.c_int, C.c_void_p, _dp, C.c_double, C.c_double, BLOCKED_FN, C.c_void_p, C.c_int, C.c_int,
                                            C.POINTER(C.c_int), _ip, _dp]
        L.frx_traj_max_rates.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.frx_objective_eval_async.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.frx_wait.argtypes = [C.c_void_p]
        L.frx_problem_set_resident.argtypes = [C.c_void_p, C.c_int]
        L.frx_optimize_path.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint)]
        L.frx_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.frx_resident_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.frx_debug_direction_log.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.frx_debug_direction_log_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.frx_debug_set_resident_retry.argtypes = [C.c_void_p, C.c_int]
        L.frx_debug_resident_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.frx_debug_resident_predictions.argtypes = [C.c_void_p, C.c_void_p]
        L.frx_debug_resident_clusters.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.frx_dilate_batch.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_void_p, C.c_double, C.c_int, _ip, _dp, _dp, _dp]
        L.frx_eval_stage_times.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.frx_eval_launch_time.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.frx_debug_set_eval_fused.argtypes = [C.c_void_p, C.c_int]
        L.frx_debug_eval_fused.argtypes = [C.c_void_p]
        L.frx_debug_set_eval_solo.argtypes = [C.c_void_p, C.c_int]
        L.frx_debug_eval_solo.argtypes = [C.c_void_p]
        L.frx_debug_penalty_kernel.argtypes = [C.c_void_p]
        L.frx_debug_mailbox_numa.argtypes = [C.c_void_p, C.c_void_p]
        L.frx_debug_profile_eval_cluster.argtypes = [C.c_void_p, _dp, C.c_void_p]
        L.frx_multi_create.argtypes = [C.POINTER(FrxConfig), C.c_int, C.c_void_p, C.c_int, _ip, _dp, _dp, _ip, _dp, _ip, _dp, C.POINTER(C.c_void_p)]
        L.frx_multi_destroy.argtypes = [C.c_void_p]
        L.frx_multi_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.frx_multi_layout.argtypes = [C.c_void_p, _ip, _ip]
        L.frx_multi_initial_guess.argtypes = [C.c_void_p, _dp]
        L.frx_multi_optimize.argtypes = [C.c_void_p, C.POINTER(LbfgsParams), _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                         _dp, _dp, C.POINTER(C.c_int)]
        L.frx_multi_last_exchange.argtypes = [C.c_void_p]
        L.frx_problem_destroy.argtypes = [C.c_void_p]
        L.frx_problem_set_solver.argtypes = [C.c_void_p, C.c_int]
        L.frx_problem_set_lbfgs_mode.argtypes = [C.c_void_p, C.c_int]
        L.frx_profile_phases.argtypes = [C.c_void_p, _dp, np.ctypeslib.ndpointer(dtype=np.int64, flags='C_CONTIGUOUS')]
        L.frx_problem_totals.argtypes = [C.c_void_p, _ip]
        L.frx_problem_layout.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip]
        L.frx_initial_guess.argtypes = [C.c_void_p, _dp]
        L.frx_objective_eval.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.frx_objective_eval_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.frx_penalty_eval.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.frx_penalty_eval_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.frx_forward.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.frx_optimize.argtypes = [C.c_void_p, C.POINTER(LbfgsParams), _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip]
        L.frx_optimize_stats.argtypes = [C.c_void_p, _dp]
        L.frx_lbfgs_minimize_batch.argtypes = [C.c_int, _ip, _dp, _dp, _ip, _ip, _ip, C.POINTER(LbfgsParams), BATCH_EVAL_FN,
                                               C.c_void_p, C.c_int]
        _vp = C.c_void_p
        L.frx_map_mark_cloud.argtypes = [C.c_double * 3, C.c_int * 3, C.c_double, C.c_int, _vp, _vp]
        L.frx_map_is_blocked.argtypes = [_vp, _vp, _vp]
        L.frx_grid_search.argtypes = [_vp, _vp, _vp, _vp, C.c_double, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
        L.frx_jps_tables.argtypes = [_vp] * 6
        L.frx_jps_plan.argtypes = [_vp, _vp, _vp, C.c_double, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.frx_route_plan.argtypes = [_vp, _vp, _vp, C.c_int, _vp, C.c_double, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
        for name in ABI_SYMBOLS + DEBUG_SYMBOLS:
            fn = getattr(L, name)
            if name not in ("frx_version", "frx_last_error", "frx_device_count", "frx_lbfgs_default_params",
                            "frx_lbfgs_gcopter_params", "frx_problem_destroy"):
                fn.restype = C.c_int
        _lib = L
    return _lib


class FrxError(RuntimeError):
    """A non-zero frx_status; `.code` is the status (include/frx.h), the text is frx_last_error()."""
    code = 0


def _check(rc: int):
    if rc != 0:
        e = FrxError(f"frx error {rc}: {lib().frx_last_error().decode()}")
        e.code = rc
        raise e


def shader_clock(device: int = 0, ms: float = 2.0):
    """frx_debug_shader_clock: (min, mean, max) MHz the device sustains under a latency-bound FP64 load."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    _check(lib().frx_debug_shader_clock(device, C.c_double(ms), C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def gcopter_lbfgs_params(rel_cost_tol: float, max_iterations: int = 0) -> LbfgsParams:
    p = LbfgsParams()
    lib().frx_lbfgs_gcopter_params(C.byref(p), rel_cost_tol)
    p.max_iterations = max_iterations
    return p


def pack_batch(cands):
    """Pack a list of scenario.Candidate into the CSR arrays of frx_problem_create."""
    coarse_n = np.array([c.coarse_n for c in cands], dtype=np.int32)
    ini = np.concatenate([c.ini_state.T.reshape(-1) for c in cands]).astype(np.float64)
    fin = np.concatenate([c.fin_state.T.reshape(-1) for c in cands]).astype(np.float64)
    h_off = [0]; v_off = [0]; h_rec = []; v_rec = []
    for c in cands:
        for h in c.h_polys:
            h_off.append(h_off[-1] + h.shape[1]); h_rec.append(h.T.reshape(-1))
        for v in c.v_polys:
            v_off.append(v_off[-1] + v.shape[1]); v_rec.append(v.T.reshape(-1))
    v_all = np.concatenate(v_rec).astype(np.float64) if v_rec else np.zeros(0)      # no vertices: frx_problem_create_from_h
    return (coarse_n, ini, fin, np.array(h_off, dtype=np.int32), np.concatenate(h_rec).astype(np.float64),
            np.array(v_off, dtype=np.int32), v_all)


def enumerate_vertices(hpoly: np.ndarray) -> np.ndarray:
    """Vertices (3 x nv) of a 6 x K H-polytope through the library (frx_enumerate_vertices)."""
    rec = np.ascontiguousarray(hpoly.T.reshape(-1), dtype=np.float64)
    nv = C.c_int()
    _check(lib().frx_enumerate_vertices(hpoly.shape[1], rec, None, 0, C.byref(nv)))
    out = np.zeros(3 * nv.value)
    _check(lib().frx_enumerate_vertices(hpoly.shape[1], rec, out.ctypes.data, nv.value, C.byref(nv)))
    return out.reshape(-1, 3).T.copy()


BLOCKED_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def line_segment_dilate(p1, p2, bbox, obs, offset: float = 0.0):
    """Corridor cell of one segment (frx_line_segment_dilate): (H 6 x K as [n; p] columns, ellipsoid C 3x3, centre d)."""
    p1 = np.ascontiguousarray(p1, dtype=np.float64); p2 = np.ascontiguousarray(p2, dtype=np.float64); bbox = np.ascontiguousarray(bbox, dtype=np.float64)
    obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
    n = C.c_int()
    op = obs.ctypes.data if len(obs) else None
    _check(lib().frx_line_segment_dilate(p1, p2, bbox, len(obs), op, offset, 0, C.byref(n), None, None, None))
    rec = np.zeros(6 * n.value); Cm = np.zeros(9); d = np.zeros(3)
    _check(lib().frx_line_segment_dilate(p1, p2, bbox, len(obs), op, offset, n.value, C.byref(n), rec.ctypes.data, Cm.ctypes.data, d.ctypes.data))
    return rec.reshape(-1, 6).T.copy(), Cm.reshape(3, 3), d


def dilate_batch(p1, p2, bbox, obs, offset: float = 0.0, cap_planes: int = 96, device: int = 0):
    """Corridor cells of a batch of segments on the device (frx_dilate_batch): list of (H 6 x K, C 3x3, d) per segment."""
    p1 = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, 3); p2 = np.ascontiguousarray(p2, dtype=np.float64).reshape(-1, 3)
    obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
    S = len(p1)
    npl = np.zeros(S, np.int32); rec = np.zeros(S * cap_planes * 6); Cm = np.zeros(S * 9); d = np.zeros(S * 3)
    _check(lib().frx_dilate_batch(device, S, p1.reshape(-1), p2.reshape(-1), np.ascontiguousarray(bbox, dtype=np.float64), len(obs),
                                  obs.ctypes.data if len(obs) else None, offset, cap_planes, npl, rec, Cm, d))
    rec = rec.reshape(S, cap_planes, 6)
    return [(rec[s, :npl[s]].T.copy(), Cm[9 * s:9 * s + 9].reshape(3, 3).copy(), d[3 * s:3 * s + 3].copy()) for s in range(S)]


def corridor_generate(path, obs, bbox, map_height: float, max_seg: float = 4.0, blocked=None, cap_polys: int = 4096, cap_planes: int = 1 << 18):
    """Greedy safe-flight corridor along `path` (n x 3) in the point cloud `obs` (frx_corridor_generate): list of 6 x K_i arrays."""
    path = np.ascontiguousarray(path, dtype=np.float64).reshape(-1, 3); obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
    user = None
    if isinstance(blocked, VoxelMap):  # MapUtil::isBlocked on the grid, without a round trip through Python
        cb = C.cast(lib().frx_map_is_blocked, BLOCKED_FN); user = C.cast(C.pointer(blocked._s), C.c_void_p)
    else:
        cb = BLOCKED_FN(lambda a, b, u: int(bool(blocked(np.array(a[:3]), np.array(b[:3]))))) if blocked else C.cast(None, BLOCKED_FN)
    n = C.c_int(); h_off = np.zeros(cap_polys + 1, dtype=np.int32); h_rec = np.zeros(6 * cap_planes)
    _check(lib().frx_corridor_generate(len(path), path.reshape(-1), len(obs), obs.ctypes.data if len(obs) else None, np.ascontiguousarray(bbox, dtype=np.float64),
                                       map_height, max_seg, cb, user, cap_polys, cap_planes, C.byref(n), h_off, h_rec))
    return [h_rec[6 * h_off[k]:6 * h_off[k + 1]].reshape(-1, 6).T.copy() for k in range(n.value)]


class VoxelMapStruct(C.Structure):
    """frx_voxel_map (include/frx.h)."""
    _fields_ = [("origin", C.c_double * 3), ("dim", C.c_int * 3), ("res", C.c_double), ("cells", C.c_void_p)]


class VoxelMap:
    """The occupancy grid of JPS::MapUtil<3> (map_util.h): origin, dim (cells per axis), res, cells[z][y][x] int8
    (0 free, 100 occupied, -1 unknown)."""

    def __init__(self, origin, dim, res: float, cells=None):
        self.origin = np.asarray(origin, dtype=np.float64).copy(); self.dim = np.asarray(dim, dtype=np.int32).copy(); self.res = float(res)
        n = int(self.dim[0]) * int(self.dim[1]) * int(self.dim[2])
        self.cells = np.zeros(n, dtype=np.int8) if cells is None else np.ascontiguousarray(cells, dtype=np.int8).reshape(-1)
        assert self.cells.size == n
        self._s = VoxelMapStruct((C.c_double * 3)(*self.origin), (C.c_int * 3)(*[int(d) for d in self.dim]), self.res, self.cells.ctypes.data)

    @classmethod
    def from_params(cls, x_size: float, y_size: float, z_size: float, res: float = 0.1):
        """MapUtil::setParam (map_util.h:48-70): origin (-x_size/2, -10, 0), dim = int(size / res)."""
        return cls([-x_size / 2, -10.0, 0.0], [int(x_size / res), int(y_size / res), int(z_size / res)], res)

    def mark_cloud(self, pts) -> int:
        """frx_map_mark_cloud: the cells holding the points become occupied; returns how many points were inside."""
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        rc = lib().frx_map_mark_cloud(self._s.origin, self._s.dim, self.res, len(pts), pts.ctypes.data if len(pts) else None, self.cells.ctypes.data)
        if rc < 0:
            _check(rc)
        return rc

    def is_blocked(self, a, b) -> bool:
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        return bool(lib().frx_map_is_blocked(a.ctypes.data, b.ctypes.data, C.byref(self._s)))

    def plan(self, start, goal, eps: float = 1.0, use_jps: bool = False, cap: int = 1 << 16):
        """frx_jps_plan -> dict(status, raw_path, path, sample_path, expanded)."""
        start = np.ascontiguousarray(start, dtype=np.float64); goal = np.ascontiguousarray(goal, dtype=np.float64)
        bufs = [np.zeros((cap, 3)) for _ in range(3)]; ns = [C.c_int() for _ in range(3)]; st = C.c_int(); ex = C.c_int()
        _check(lib().frx_jps_plan(C.byref(self._s), start.ctypes.data, goal.ctypes.data, float(eps), int(use_jps), cap,
                                  C.byref(ns[0]), bufs[0].ctypes.data, C.byref(ns[1]), bufs[1].ctypes.data, C.byref(ns[2]), bufs[2].ctypes.data,
                                  C.byref(st), C.byref(ex)))
        return dict(status=st.value, raw_path=bufs[0][:ns[0].value].copy(), path=bufs[1][:ns[1].value].copy(),
                    sample_path=bufs[2][:ns[2].value].copy(), expanded=ex.value)

    def route(self, start, goal, gates=(), eps: float = 1.0, use_jps: bool = False, threads: int = 0, cap: int = 1 << 18):
        """frx_route_plan -> (path n x 3 (empty when a leg failed), leg_status, leg_expanded)."""
        start = np.ascontiguousarray(start, dtype=np.float64); goal = np.ascontiguousarray(goal, dtype=np.float64)
        gates = np.ascontiguousarray(gates, dtype=np.float64).reshape(-1, 3)
        out = np.zeros((cap, 3)); n = C.c_int(); st = np.zeros(len(gates) + 1, dtype=np.int32); ex = np.zeros(len(gates) + 1, dtype=np.int32)
        _check(lib().frx_route_plan(C.byref(self._s), start.ctypes.data, goal.ctypes.data, len(gates), gates.ctypes.data if len(gates) else None,
                                    float(eps), int(use_jps), int(threads), cap, C.byref(n), out.ctypes.data, st.ctypes.data, ex.ctypes.data))
        return out[:n.value].copy(), st, ex


def grid_search(cmap, dim, start, goal, eps: float = 1.0, use_jps: bool = False, max_expand: int = -1, cap: int = 1 << 20):
    """frx_grid_search on an int8 occupancy array (x fastest; dim[2] = 0 for 2-D): (path goal-first k x 3 int, expanded, cost)."""
    cmap = np.ascontiguousarray(cmap, dtype=np.int8).reshape(-1)
    dim = np.asarray(dim, dtype=np.int32); start = np.asarray(start, dtype=np.int32); goal = np.asarray(goal, dtype=np.int32)
    path = np.zeros((cap, 3), dtype=np.int32); n = C.c_int(); ex = C.c_int(); cost = C.c_double()
    _check(lib().frx_grid_search(cmap.ctypes.data, dim.ctypes.data, start.ctypes.data, goal.ctypes.data, float(eps), int(use_jps), int(max_expand),
                                 cap, C.byref(n), path.ctypes.data, C.byref(ex), C.byref(cost)))
    return path[:n.value].copy(), ex.value, cost.value


def jps_tables():
    """The jump-point neighbour tables in the reference's storage order (frx_jps_tables)."""
    t = [np.zeros(s, dtype=np.int32) for s in ((27, 3, 26), (27, 3, 12), (27, 3, 12), (9, 2, 8), (9, 2, 2), (9, 2, 2))]
    _check(lib().frx_jps_tables(*[a.ctypes.data for a in t]))
    return t


def dv_selftest(n, B=4, m=128, iters=140, geom=None, seed=0, device=0):
    """(worst relative error of the device search direction vs a host two-loop recursion, mean us of the last launches)."""
    err, us = C.c_double(), C.c_double()
    gp = (C.c_int * 4)(*geom) if geom else None
    _check(lib().frx_dv_selftest(device, n, B, m, iters, gp, seed, C.byref(err), C.byref(us)))
    return err.value, us.value


def traj_max_rates(T, Cf):
    """(max |v|, max |a|) per piece (frx_traj_max_rates)."""
    n = len(T); mv = np.zeros(n); ma = np.zeros(n)
    _check(lib().frx_traj_max_rates(n, np.ascontiguousarray(T, dtype=np.float64), np.ascontiguousarray(Cf, dtype=np.float64).reshape(-1), mv, ma))
    return mv, ma


def traj_to_msg(T, Cf):
    """PolynomialTrajectory array fields of one trajectory: (coef_x, coef_y, coef_z, time, order)."""
    n = len(T)
    cx = np.zeros(6 * n); cy = np.zeros(6 * n); cz = np.zeros(6 * n); tm = np.zeros(n); od = np.zeros(n, np.uint32)
    _check(lib().frx_traj_to_msg(n, np.ascontiguousarray(T, dtype=np.float64), np.ascontiguousarray(Cf, dtype=np.float64).reshape(-1), cx, cy, cz, tm, od))
    return cx, cy, cz, tm, od


def msg_sample(msg, t: float):
    cx, cy, cz, tm, od = msg
    p = np.zeros(3); v = np.zeros(3); a = np.zeros(3); j = np.zeros(3)
    _check(lib().frx_msg_sample(len(tm), cx, cy, cz, tm, od, float(t), p, v, a, j))
    return p, v, a, j


class MultiProblem:
    """A batch sharded over several devices behind the C ABI (frx_multi_*): one host thread and handle per device, RCCL winner exchange."""

    def __init__(self, cands, params: dict, devices=None, n_devices: int = 0, **override):
        self.cfg = FrxConfig.from_params(params, **override)
        coarse_n, ini, fin, h_off, h_rec, v_off, v_rec = pack_batch(cands)
        h = C.c_void_p()
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        nd = n_devices if devices is None else len(devices)
        _check(lib().frx_multi_create(C.byref(self.cfg), nd, None if dev is None else dev.ctypes.data, len(cands), coarse_n, ini, fin, h_off, h_rec, v_off, v_rec, C.byref(h)))
        self.h = h
        self.B = len(cands)
        g = C.c_int(); r = C.c_int()
        _check(lib().frx_multi_info(self.h, C.byref(g), C.byref(r), None, None))
        self.n_shards, self.uses_rccl = int(g.value), bool(r.value)
        self.shard_lo = np.zeros(self.n_shards + 1, np.int32); self.shard_device = np.zeros(self.n_shards, np.int32)
        _check(lib().frx_multi_info(self.h, None, None, self.shard_lo.ctypes.data, self.shard_device.ctypes.data))
        self.piece_off = np.zeros(self.B + 1, np.int32); self.x_off = np.zeros(self.B + 1, np.int32)
        _check(lib().frx_multi_layout(self.h, self.piece_off, self.x_off))
        self.P, self.NX = int(self.piece_off[-1]), int(self.x_off[-1])
        self.maxN = int(np.diff(self.piece_off).max())

    def close(self):
        if getattr(self, "h", None):
            lib().frx_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def initial_guess(self):
        x = np.zeros(self.NX)
        _check(lib().frx_multi_initial_guess(self.h, x))
        return x

    def optimize(self, rel_cost_tol: float, x0=None, max_iterations: int = 0):
        x = self.initial_guess() if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        pm = gcopter_lbfgs_params(rel_cost_tol, max_iterations)
        Cf = np.zeros(self.P * 18); T = np.zeros(self.P); jc = np.zeros(self.B); obj = np.zeros(self.B)
        st = np.zeros(self.B, np.int32); it = np.zeros(self.B, np.int32); ev = np.zeros(self.B, np.int32)
        wid = C.c_int(); wobj = C.c_double(); wn = C.c_int()
        wC = np.zeros(self.maxN * 18); wT = np.zeros(self.maxN)
        _check(lib().frx_multi_optimize(self.h, C.byref(pm), x, Cf, T, jc, obj, st, it, ev, C.byref(wid), C.byref(wobj), wC, wT, C.byref(wn)))
        n = int(wn.value)
        return dict(x=x, C=Cf.reshape(-1, 3), T=T, jerk_cost=jc, objective=obj, status=st, iters=it, evals=ev, winner_id=int(wid.value),
                    winner_objective=float(wobj.value), winner_C=wC[:18 * n].reshape(-1, 3).copy(), winner_T=wT[:n].copy(),
                    exchange="rccl" if lib().frx_multi_last_exchange(self.h) else "host")


class Problem:
    """A batch of candidate trajectories resident on one MI355X: the SE3GCOPTER::setup / optimize
    pair (CPU.hpp:1076, :1230) behind the C ABI."""

    def __init__(self, cands, params: dict, device: int = 0, enumerate_v: bool = False, packed=None, **override):
        self.cfg = FrxConfig.from_params(params, **override)
        self.kappa = int(self.cfg.qd_intervals)
        coarse_n, ini, fin, h_off, h_rec, v_off, v_rec = packed if packed is not None else pack_batch(cands)   # packed: pack_batch(cands) done by the caller (timing)
        h = C.c_void_p()
        if enumerate_v:      # V-polytopes from the library's own H->V enumeration (frx_problem_create_from_h)
            _check(lib().frx_problem_create_from_h(C.byref(self.cfg), device, len(cands), coarse_n, ini, fin, h_off, h_rec, C.byref(h)))
        else:
            _check(lib().frx_problem_create(C.byref(self.cfg), device, len(cands), coarse_n, ini, fin, h_off, h_rec, v_off, v_rec,
                                            C.byref(h)))
        self.h = h
        t = np.zeros(6, dtype=np.int32)
        _check(lib().frx_problem_totals(self.h, t))
        self.B, self.P, self.Pc, self.NX, self.Kmax, self.sum_K = (int(v) for v in t)
        self.piece_off = np.zeros(self.B + 1, np.int32); self.coarse_off = np.zeros(self.B + 1, np.int32)
        self.x_off = np.zeros(self.B + 1, np.int32); self.dim_t = np.zeros(self.B, np.int32)
        _check(lib().frx_problem_layout(self.h, self.piece_off, self.coarse_off, self.x_off, self.dim_t))

    def close(self):
        if getattr(self, "h", None):
            lib().frx_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: module globals may already be gone
            pass

    def set_solver(self, name: str):
        """'knot_pcr' (default) or 'banded_lu' (reference elimination order, cross-check)."""
        _check(lib().frx_problem_set_solver(self.h, {"knot_pcr": 0, "banded_lu": 1}[name]))

    def set_lbfgs_mode(self, name: str):
        """'device' (default: vectors on the GPU, decisions on the host) or 'host' (reference-exact host vectors)."""
        _check(lib().frx_problem_set_lbfgs_mode(self.h, {"device": 0, "host": 1}[name]))

    def set_resident(self, enable):
        """0 / False: one launch per stage and round; 1 / True: resident round kernel when the batch fits the chip, its work queue for batches up
        to a few times that size (default); 2: the work queue for every batch that exceeds the chip."""
        _check(lib().frx_problem_set_resident(self.h, int(enable)))

    def resident_clusters(self):
        """Clusters of the last resident plan (< B: the candidates went through the work queue)."""
        a = C.c_int()
        _check(lib().frx_debug_resident_clusters(self.h, C.byref(a)))
        return int(a.value)

    def optimize_path(self):
        """(resident_used, device_status) of the last optimize()."""
        u = C.c_int(); st = C.c_uint()
        _check(lib().frx_optimize_path(self.h, C.byref(u), C.byref(st)))
        return int(u.value), int(st.value)

    def trace(self):
        """Rows {flags, step, f, g.d, gp.d_new, x.x, g.g} of candidate 0's evaluated commands (needs FRX_TRACE in the environment)."""
        n = lib().frx_debug_trace(self.h, None, 0)
        out = np.zeros((max(n, 0), 7))
        if n > 0:
            lib().frx_debug_trace(self.h, out.ctypes.data, n)
        return out

    def direction_log(self, cap_steps: int, n_cands: int = 1):
        """Record (s, y, g, d) of every accepted step of later resident plans (frx_debug_direction_log); 0 switches it off."""
        _check(lib().frx_debug_direction_log(self.h, int(cap_steps), int(n_cands)))

    def read_direction_log(self, cand: int = 0):
        """dict(s, y, g, d: rows x nxp arrays; slot, bound: rows) of candidate `cand`'s accepted steps in the last resident plan."""
        rows = C.c_int(); rd = C.c_int()
        _check(lib().frx_debug_direction_log_read(self.h, cand, None, 0, C.byref(rows), C.byref(rd)))
        out = np.zeros((rows.value, rd.value))
        if rows.value:
            _check(lib().frx_debug_direction_log_read(self.h, cand, out.ctypes.data, rows.value, C.byref(rows), C.byref(rd)))
        nxp = (rd.value - 2) // 4
        return dict(s=out[:, :nxp], y=out[:, nxp:2 * nxp], g=out[:, 2 * nxp:3 * nxp], d=out[:, 3 * nxp:4 * nxp],
                    slot=out[:, 4 * nxp].astype(int), bound=out[:, 4 * nxp + 1].astype(int))

    def set_resident_retry(self, enable: bool):
        """Diagnostic: re-run candidates that fail on the resident kernel on the per-stage rounds (round-2 behaviour; default off)."""
        _check(lib().frx_debug_set_resident_retry(self.h, 1 if enable else 0))

    def set_takeover_at(self, rounds: int):
        """Tests (frx_debug_set_takeover_at): rounds > 0 = every plan starts as per-stage rounds and hands its candidates to the resident kernel after so many."""
        _check(lib().frx_debug_set_takeover_at(self.h, C.c_long(int(rounds))))

    def eval_status(self):
        """frx_eval_status: has a wait inside a one-launch evaluation expired since the last check?  Raises FrxError (FRX_ERR_TIMEOUT) if so; the caller's stream must be synchronised."""
        _check(lib().frx_eval_status(self.h))

    def taken_over(self) -> int:
        """Candidates of the last plan that began as per-stage rounds and finished on the resident round kernel (frx_debug_taken_over)."""
        n = C.c_int(0)
        _check(lib().frx_debug_taken_over(self.h, C.byref(n)))
        return int(n.value)

    def resident_predictions(self):
        """(rounds on a predicted ADVANCE, rounds on a predicted trial step, predictions redone) of the last resident plan, all candidates."""
        out = np.zeros(3, dtype=np.uint64)
        _check(lib().frx_debug_resident_predictions(self.h, out.ctypes.data))
        return tuple(int(v) for v in out)

    def resident_counts(self):
        """(failed, retried) candidates of the last resident plan."""
        a = C.c_int(); b = C.c_int()
        _check(lib().frx_debug_resident_counts(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def resident_profile(self):
        """[S][G][16] microseconds per segment of the last resident plan, S clusters (needs FRX_RESIDENT_PROF in the environment)."""
        n = lib().frx_resident_profile(self.h, None, 0)
        if n <= 0:
            return None
        out = np.zeros(n, dtype=np.uint64)
        lib().frx_resident_profile(self.h, out.ctypes.data, n)
        self.last_stamps = out[-32:].astype(np.int64)                  # shader-clock stamps of candidate 0's forward (0..6) / adjoint (16..24) bodies
        body = out[:-32]
        S = self.resident_clusters()
        self.last_host_wait_hist = body[-16 * S:].reshape(S, 16).astype(np.int64)   # per leader: waits for a host command, bin k = shorter than 2^k us
        return body[:-16 * S].reshape(S, -1, 16).astype(np.float64) / 100.0

    def initial_guess(self):
        x = np.zeros(self.NX)
        _check(lib().frx_initial_guess(self.h, x))
        return x

    def objective(self, x):
        f = np.zeros(self.B); g = np.zeros(self.NX)
        _check(lib().frx_objective_eval(self.h, np.ascontiguousarray(x, dtype=np.float64), f, g))
        return f, g

    def objective_device(self, x_ptr: int, f_ptr: int, g_ptr: int, stream: int = 0):
        _check(lib().frx_objective_eval_device(self.h, x_ptr, f_ptr, g_ptr, stream))

    def penalty(self, T, Cf):
        cost = np.zeros(self.B); gdT = np.zeros(self.P); gdC = np.zeros(self.P * 18)
        _check(lib().frx_penalty_eval(self.h, np.ascontiguousarray(T, dtype=np.float64),
                                      np.ascontiguousarray(Cf, dtype=np.float64).reshape(-1), cost, gdT, gdC))
        return cost, gdT, gdC.reshape(-1, 3)

    def penalty_device(self, T_ptr: int, C_ptr: int, out_ptr: int, stream: int = 0):
        _check(lib().frx_penalty_eval_device(self.h, T_ptr, C_ptr, out_ptr, stream))

    def forward(self, x):
        T = np.zeros(self.P); Cf = np.zeros(self.P * 18)
        _check(lib().frx_forward(self.h, np.ascontiguousarray(x, dtype=np.float64), T, Cf))
        return T, Cf.reshape(-1, 3)

    def optimize(self, rel_cost_tol: float, x0=None, max_iterations: int = 0):
        x = self.initial_guess() if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        pm = gcopter_lbfgs_params(rel_cost_tol, max_iterations)
        Cf = np.zeros(self.P * 18); T = np.zeros(self.P)
        jc = np.zeros(self.B); obj = np.zeros(self.B)
        st = np.zeros(self.B, np.int32); it = np.zeros(self.B, np.int32); ev = np.zeros(self.B, np.int32)
        _check(lib().frx_optimize(self.h, C.byref(pm), x, Cf, T, jc, obj, st, it, ev))
        stats = np.zeros(4)
        _check(lib().frx_optimize_stats(self.h, stats))
        resident, dev_status = self.optimize_path()
        return dict(x=x, C=Cf.reshape(-1, 3), T=T, jerk_cost=jc, objective=obj, status=st, iters=it, evals=ev,
                    ms_total=stats[0], ms_device=stats[1], ms_host=stats[2], rounds=int(stats[3]), resident=resident, device_status=dev_status,
                    resident_failed=self.resident_counts()[0], resident_retried=self.resident_counts()[1], predictions=self.resident_predictions(),
                    clusters=self.resident_clusters() if resident else 0, taken_over=self.taken_over())

    def stage_times(self, x, reps: int = 100):
        """Average microseconds of the forward, penalty and adjoint kernels at x (HIP events inside the library)."""
        out = np.zeros(3)
        _check(lib().frx_eval_stage_times(self.h, np.ascontiguousarray(x, dtype=np.float64), reps, out))
        return {"forward": float(out[0]), "penalty": float(out[1]), "adjoint": float(out[2])}

    def eval_launch_time(self, x, reps: int = 100) -> float:
        """Average microseconds of one evaluation at x in the form frx_objective_eval_device takes (HIP events inside the library)."""
        out = np.zeros(1)
        _check(lib().frx_eval_launch_time(self.h, np.ascontiguousarray(x, dtype=np.float64), reps, out))
        return float(out[0])

    def profile_eval_cluster(self, x):
        """Shader-clock stamps of cluster 0 during one evaluation in the one-launch form (frx_debug.h: frx_debug_profile_eval_cluster)."""
        out = np.zeros(64, np.int64)
        _check(lib().frx_debug_profile_eval_cluster(self.h, np.ascontiguousarray(x, dtype=np.float64), out.ctypes.data))
        return out

    def set_eval_fused(self, on: bool):
        """Diagnostic: False = evaluations as three stage launches, True = the default (one launch where it applies, frx_eval_kernel.hpp)."""
        _check(lib().frx_debug_set_eval_fused(self.h, int(on)))          # (2: test mode - every wait inside the launch expires at once)

    def eval_fused(self) -> int:
        """Workgroups per candidate of the one-launch evaluation in use, 0 = one launch per stage."""
        return int(lib().frx_debug_eval_fused(self.h))

    def set_eval_solo(self, mode: int):
        """Diagnostic: 0 = never the solo form (one workgroup per candidate, one launch per evaluation, frx_solo_kernel.hpp), 1 = from the handle's batch-size
        threshold on (default), 2 = at every batch size."""
        _check(lib().frx_debug_set_eval_solo(self.h, int(mode)))

    def solo_applies(self) -> bool:
        """Does the solo form exist for this handle's geometry (<= 64 pieces and samples per piece, knot solver)?"""
        was = int(lib().frx_debug_set_eval_solo(self.h, 2))
        if was != 0: return False
        ok = self.eval_solo() >= 1
        _check(lib().frx_debug_set_eval_solo(self.h, 1))
        return ok

    def eval_solo(self) -> int:
        """Workgroups of the solo kernel a CU holds if the next evaluation takes that form, 0 = it does not."""
        return int(lib().frx_debug_eval_solo(self.h))

    def mailbox_numa(self):
        """{NUMA node of the resident plan's command / result mailbox pages, the device's node, the caller's CPU}; -1 = unknown."""
        out = np.zeros(4, np.int32)
        _check(lib().frx_debug_mailbox_numa(self.h, out.ctypes.data))
        return {"cmd_node": int(out[0]), "res_node": int(out[1]), "device_node": int(out[2]), "caller_cpu": int(out[3])}

    def penalty_kernel(self) -> str:
        """Name of the penalty kernel a stage launch of this handle takes."""
        return {0: "frx::k_penalty", 1: "frx::k_penalty_lat", 2: "frx::k_penalty_lat2"}.get(int(lib().frx_debug_penalty_kernel(self.h)), "?")

    def algorithmic_bytes(self) -> int:
        """Penalty-kernel bytes per evaluation, SURVEY.md §8d: sum over pieces of 312 + 48 K_i."""
        return 312 * self.P + 48 * self.sum_K

    def samples(self) -> int:
        """constraint samples per evaluation: pieces x (kappa + 1)."""
        return self.P * (self.kappa + 1)


class PenaltyProblem(Problem):
    """The inner boundary on its own (frx_penalty_problem_create): a handle built from what cuda_computer::compute receives on every call - per piece the index of
    its H-polytope (idxHs), the polytopes (cfgHs), the scalar limits and weights - serving `penalty` / `penalty_device` only (cuda_computer.cuh:118-134)."""

    def __init__(self, params: dict, piece_n, piece_poly, h_polys, device: int = 0, **override):
        self.cfg = FrxConfig.from_params(params, **override)
        self.kappa = int(self.cfg.qd_intervals)
        piece_n = np.ascontiguousarray(piece_n, dtype=np.int32); piece_poly = np.ascontiguousarray(piece_poly, dtype=np.int32)
        h_off = np.zeros(len(h_polys) + 1, dtype=np.int32)
        for i, hp in enumerate(h_polys):
            h_off[i + 1] = h_off[i] + hp.shape[1]
        h_rec = np.ascontiguousarray(np.concatenate([hp.T.reshape(-1) for hp in h_polys]), dtype=np.float64)
        h = C.c_void_p()
        _check(lib().frx_penalty_problem_create(C.byref(self.cfg), device, len(piece_n), C.c_void_p(piece_n.ctypes.data), C.c_void_p(piece_poly.ctypes.data),
                                                C.c_void_p(h_off.ctypes.data), C.c_void_p(h_rec.ctypes.data), C.byref(h)))
        self.h = h
        t = np.zeros(6, dtype=np.int32)
        _check(lib().frx_problem_totals(self.h, t))
        self.B, self.P, self.Pc, self.NX, self.Kmax, self.sum_K = (int(v) for v in t)
        self.piece_off = np.zeros(self.B + 1, np.int32); self.coarse_off = np.zeros(self.B + 1, np.int32)
        self.x_off = np.zeros(self.B + 1, np.int32); self.dim_t = np.zeros(self.B, np.int32)
        _check(lib().frx_problem_layout(self.h, self.piece_off, self.coarse_off, self.x_off, self.dim_t))
