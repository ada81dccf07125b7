"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/liboracle.so) and of the
reference-derived checkers under oracle/_ref/.  Import only from tests/, smoke() and the
cpu_baseline leg of bench.py."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

EVAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)


class Config(C.Structure):
    """Layout shared by orc::Config (oracle) and frx_config (include/frx.h)."""
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


def lbfgs_params(mem_size=8, g_epsilon=1e-5, past=0, delta=1e-5, max_iterations=0, max_linesearch=40,
                 min_step=1e-20, max_step=1e20, f_dec_coeff=1e-4, s_curv_coeff=0.9, xtol=1e-16):
    return np.array([mem_size, g_epsilon, past, delta, max_iterations, max_linesearch, min_step, max_step,
                     f_dec_coeff, s_curv_coeff, xtol], dtype=np.float64)


_lib = None
_ref_lbfgs = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing — run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Config), _dp, _dp, C.c_int, _ip, _dp, _ip, _dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_dims.argtypes = [C.c_void_p, _ip]
        L.orc_set_abscissa_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_maps.argtypes = [C.c_void_p, _ip, _ip, _ip]
        L.orc_initial_guess.argtypes = [C.c_void_p, _dp]
        L.orc_set_initial.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_objective.restype = C.c_double
        L.orc_objective.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_forward.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.orc_backward.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_generate.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_dense_A.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_solve_adj.argtypes = [C.c_void_p, _dp]
        L.orc_solve.argtypes = [C.c_void_p, _dp]
        L.orc_backprop.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_jerk_cost.restype = C.c_double
        L.orc_jerk_cost.argtypes = [C.c_void_p]
        L.orc_penalty.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.orc_optimize.restype = C.c_int
        L.orc_optimize.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, C.c_int, _dp, _dp, C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_int)]
        L.orc_trace_begin.argtypes = [C.c_void_p, C.c_long]
        L.orc_trace_get.restype = C.c_long
        L.orc_trace_get.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_set_lbfgs.argtypes = [C.c_void_p]
        L.orc_lbfgs_run.restype = C.c_int
        L.orc_lbfgs_run.argtypes = [C.c_int, _dp, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, _dp, C.c_int, _dp, _dp, _ip,
                                    C.POINTER(C.c_int)]
        L.orc_objective_fnptr.restype = C.c_void_p
        _lib = L
    return _lib


def ref_lbfgs():
    """oracle/_ref/libref_lbfgs.so = the reference's own lbfgs.hpp compiled where it lies; None if absent."""
    global _ref_lbfgs
    if _ref_lbfgs is None:
        path = os.path.join(HERE, "_ref", "libref_lbfgs.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_lbfgs_run.restype = C.c_int
        R.ref_lbfgs_run.argtypes = [C.c_int, _dp, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, _dp, C.c_int, _dp, _dp, _ip,
                                    C.POINTER(C.c_int)]
        R.ref_lbfgs_optimize.restype = C.c_int
        _ref_lbfgs = R
    return _ref_lbfgs


_ref_gcopter = None


def ref_gcopter():
    """oracle/_ref/libref_gcopter.so = the reference's own CPU path (se3gcopter_cpu.hpp, trajectory.hpp, geoutils.hpp,
    sdlp.hpp, quickhull.hpp, lbfgs.hpp) compiled where it lies against oracle/eigen_shim; None if absent."""
    global _ref_gcopter
    if _ref_gcopter is None:
        path = os.path.join(HERE, "_ref", "libref_gcopter.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_create.restype = C.c_void_p
        R.ref_create.argtypes = [C.POINTER(Config), _dp, _dp, C.c_int, _ip, _dp, _ip, _dp, C.c_int]
        R.ref_destroy.argtypes = [C.c_void_p]
        R.ref_dims.argtypes = [C.c_void_p, _ip]
        R.ref_vpoly.restype = C.c_int
        R.ref_vpoly.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        R.ref_initial_guess.argtypes = [C.c_void_p, _dp]
        R.ref_objective.restype = C.c_double
        R.ref_objective.argtypes = [C.c_void_p, _dp, _dp]
        R.ref_forward.argtypes = [C.c_void_p, _dp, _dp, _dp]
        R.ref_penalty.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        R.ref_optimize.restype = C.c_double
        R.ref_optimize.argtypes = [C.c_void_p, C.c_double, _dp, _dp]
        _ref_gcopter = R
    return _ref_gcopter


_ref_gcopter_gpu = None


class _Renamed:
    """libref_gcopter_gpu.so exports refgpu_*; this view answers to the ref_* names Reference uses."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        return getattr(self._lib, "refgpu_" + name[4:] if name.startswith("ref_") else name)


def ref_gcopter_gpu():
    """oracle/_ref/libref_gcopter_gpu.so = the reference's GPU-flavoured header (se3gcopter_gpu.hpp) compiled UNMODIFIED where it lies, its `class cuda_computer`
    supplied by oracle/frx_dropin/cuda_computer.cuh, i.e. by libfrx.so (needs a HIP device at run time: the library has no CPU fallback); None if absent."""
    global _ref_gcopter_gpu
    if _ref_gcopter_gpu is None:
        path = os.path.join(HERE, "_ref", "libref_gcopter_gpu.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.refgpu_create.restype = C.c_void_p
        R.refgpu_create.argtypes = [C.POINTER(Config), _dp, _dp, C.c_int, _ip, _dp, _ip, _dp, C.c_int]
        R.refgpu_destroy.argtypes = [C.c_void_p]
        R.refgpu_dims.argtypes = [C.c_void_p, _ip]
        R.refgpu_vpoly.restype = C.c_int
        R.refgpu_vpoly.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        R.refgpu_initial_guess.argtypes = [C.c_void_p, _dp]
        R.refgpu_objective.restype = C.c_double
        R.refgpu_objective.argtypes = [C.c_void_p, _dp, _dp]
        R.refgpu_forward.argtypes = [C.c_void_p, _dp, _dp, _dp]
        R.refgpu_penalty.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        R.refgpu_optimize.restype = C.c_double
        R.refgpu_optimize.argtypes = [C.c_void_p, C.c_double, _dp, _dp]
        R.refgpu_compute_calls.restype = C.c_long
        R.refgpu_compute_calls.argtypes = [C.c_void_p]
        R.refgpu_kill_kernel.argtypes = [C.c_void_p]
        _ref_gcopter_gpu = _Renamed(R)
    return _ref_gcopter_gpu


class Reference:
    """The reference's SE3GCOPTER itself (compiled from /root/reference against the Eigen shim).  gpu_flavour: se3gcopter_gpu.hpp with the libfrx-backed
    drop-in for its cuda_computer (ref_gcopter_gpu above) instead of se3gcopter_cpu.hpp."""

    def __init__(self, cand, params: dict, override_vs: bool = True, gpu_flavour: bool = False, **override):
        R = ref_gcopter_gpu() if gpu_flavour else ref_gcopter()
        if R is None:
            raise RuntimeError("oracle/_ref/libref_gcopter%s.so not built" % ("_gpu" if gpu_flavour else ""))
        self.R = R
        self.cfg = Config.from_params(params, **override)
        h_off, h_rec, v_off, v_rec = cand.packed()
        ini = np.ascontiguousarray(cand.ini_state.T.reshape(-1)); fin = np.ascontiguousarray(cand.fin_state.T.reshape(-1))
        self.h = R.ref_create(C.byref(self.cfg), ini, fin, cand.coarse_n, h_off, h_rec, v_off, v_rec, int(override_vs))
        if not self.h:
            raise RuntimeError("reference setup failed (empty polytope, or vertex counts differ from the supplied V-polytopes)")
        d = np.zeros(4, dtype=np.int32)
        R.ref_dims(self.h, d)
        self.coarse_n, self.fine_n, self.dim_t, self.dim_p = (int(v) for v in d)
        self.n = self.dim_t + self.dim_p

    def __del__(self):
        if getattr(self, "h", None):
            self.R.ref_destroy(self.h); self.h = None

    def vpoly(self, m):
        nv = self.R.ref_vpoly(self.h, m, None, 0)
        out = np.zeros(3 * nv)
        self.R.ref_vpoly(self.h, m, out.ctypes.data, nv)
        return out.reshape(nv, 3).T.copy()

    def initial_guess(self):
        x = np.zeros(self.n); self.R.ref_initial_guess(self.h, x); return x

    def objective(self, x):
        g = np.zeros(self.n)
        f = self.R.ref_objective(self.h, np.ascontiguousarray(x, dtype=np.float64), g)
        return f, g

    def forward(self, x):
        T = np.zeros(self.fine_n); Cf = np.zeros(18 * self.fine_n)
        self.R.ref_forward(self.h, np.ascontiguousarray(x, dtype=np.float64), T, Cf)
        return T, Cf.reshape(-1, 3)

    def penalty(self, T, Cf):
        cost = np.zeros(1); gdT = np.zeros(self.fine_n); gdC = np.zeros(18 * self.fine_n)
        self.R.ref_penalty(self.h, np.ascontiguousarray(T, dtype=np.float64), np.ascontiguousarray(Cf, dtype=np.float64).reshape(-1), cost, gdT, gdC)
        return float(cost[0]), gdT, gdC.reshape(-1, 3)

    def optimize(self, rel_cost_tol):
        Cf = np.zeros(18 * self.fine_n); T = np.zeros(self.fine_n)
        jc = self.R.ref_optimize(self.h, rel_cost_tol, Cf, T)
        return dict(jerk_cost=jc, C=Cf.reshape(-1, 3), T=T)

    def compute_calls(self):
        """GPU flavour: calls of cuda_computer::compute served so far."""
        return int(self.R.refgpu_compute_calls(self.h))

    def kill_kernel(self):
        """GPU flavour: SE3GCOPTER::kill_kernel (se3gcopter_gpu.hpp:907-909)."""
        self.R.refgpu_kill_kernel(self.h)


def use_reference_lbfgs(enable: bool) -> bool:
    """Make the oracle's optimize()/backwardP run on the reference's own L-BFGS (oracle/_ref)."""
    R = ref_lbfgs() if enable else None
    if enable and R is None:
        return False
    lib().orc_set_lbfgs(C.cast(R.ref_lbfgs_optimize, C.c_void_p) if enable else None)
    return True


class Oracle:
    """One trajectory problem = SE3GCOPTER after setup() (CPU.hpp:1076-1186)."""

    def __init__(self, cand, params: dict, **override):
        self.cfg = Config.from_params(params, **override)
        h_off, h_rec, v_off, v_rec = cand.packed()
        ini = np.ascontiguousarray(cand.ini_state.T.reshape(-1))   # column-major 3x3
        fin = np.ascontiguousarray(cand.fin_state.T.reshape(-1))
        self.h = lib().orc_create(C.byref(self.cfg), ini, fin, cand.coarse_n, h_off, h_rec, v_off, v_rec)
        if not self.h:
            raise RuntimeError("oracle setup failed (empty polytope interior)")
        d = np.zeros(4, dtype=np.int32)
        lib().orc_dims(self.h, d)
        self.coarse_n, self.fine_n, self.dim_t, self.dim_p = (int(v) for v in d)
        self.n = self.dim_t + self.dim_p

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def set_abscissa_mode(self, accumulate: bool):
        lib().orc_set_abscissa_mode(self.h, int(accumulate))

    def maps(self):
        iv = np.zeros(self.coarse_n, np.int32); ivs = np.zeros(max(self.fine_n - 1, 1), np.int32); ihs = np.zeros(self.fine_n, np.int32)
        lib().orc_get_maps(self.h, iv, ivs, ihs)
        return iv, ivs[: self.fine_n - 1], ihs

    def initial_guess(self):
        x = np.zeros(self.n)
        lib().orc_initial_guess(self.h, x)
        return x

    def set_initial(self):
        T = np.zeros(self.coarse_n); P = np.zeros(3 * (self.fine_n - 1))
        lib().orc_set_initial(self.h, T, P)
        return T, P.reshape(-1, 3)

    def objective(self, x):
        g = np.zeros(self.n)
        f = lib().orc_objective(self.h, np.ascontiguousarray(x, dtype=np.float64), g)
        return f, g

    def forward(self, x):
        T = np.zeros(self.fine_n); P = np.zeros(3 * (self.fine_n - 1)); Cf = np.zeros(18 * self.fine_n)
        lib().orc_forward(self.h, np.ascontiguousarray(x, dtype=np.float64), T, P, Cf)
        return T, P.reshape(-1, 3), Cf.reshape(-1, 3)

    def backward(self, Tcoarse, P):
        x = np.zeros(self.n)
        lib().orc_backward(self.h, np.ascontiguousarray(Tcoarse, dtype=np.float64), np.ascontiguousarray(P, dtype=np.float64).reshape(-1), x)
        return x

    def generate(self, inPs, T):
        Cf = np.zeros(18 * self.fine_n)
        lib().orc_generate(self.h, np.ascontiguousarray(inPs, dtype=np.float64).reshape(-1), np.ascontiguousarray(T, dtype=np.float64), Cf)
        return Cf.reshape(-1, 3)

    def dense_A(self, T):
        n6 = 6 * self.fine_n
        A = np.zeros(n6 * n6)
        lib().orc_dense_A(self.h, np.ascontiguousarray(T, dtype=np.float64), A)
        return A.reshape(n6, n6)

    def solve_adj(self, rhs):
        r = np.ascontiguousarray(rhs, dtype=np.float64).reshape(-1).copy()
        lib().orc_solve_adj(self.h, r)
        return r.reshape(-1, 3)

    def solve(self, rhs):
        r = np.ascontiguousarray(rhs, dtype=np.float64).reshape(-1).copy()
        lib().orc_solve(self.h, r)
        return r.reshape(-1, 3)

    def backprop(self, gdC):
        """lambda = A^-T gdC with the factors of the last generate(); returns (gdT[N], gdP[N-1,3]) from zero."""
        lam = np.ascontiguousarray(gdC, dtype=np.float64).reshape(-1).copy()
        gdT = np.zeros(self.fine_n); gdP = np.zeros(3 * (self.fine_n - 1))
        lib().orc_backprop(self.h, lam, gdT, gdP)
        return gdT, gdP.reshape(-1, 3)

    def jerk_cost(self):
        return lib().orc_jerk_cost(self.h)

    def penalty(self, T, Cf):
        """a7 alone: returns (cost, gdT[N], gdC[6N,3]) accumulated from zero."""
        cost = np.zeros(1); gdT = np.zeros(self.fine_n); gdC = np.zeros(18 * self.fine_n)
        lib().orc_penalty(self.h, np.ascontiguousarray(T, dtype=np.float64), np.ascontiguousarray(Cf, dtype=np.float64).reshape(-1), cost, gdT, gdC)
        return float(cost[0]), gdT, gdC.reshape(-1, 3)

    def optimize_traced(self, rel_cost_tol, cap=20000, **kw):
        """optimize() that also returns every point the solver evaluated, in order (rows of an array)."""
        lib().orc_trace_begin(self.h, cap)
        r = self.optimize(rel_cost_tol, **kw)
        buf = np.zeros((cap, self.n))
        cnt = lib().orc_trace_get(self.h, buf.ctypes.data, cap)
        r["trace"] = buf[:cnt].copy()
        return r

    def optimize(self, rel_cost_tol, max_iterations=0, x0=None):
        x = np.zeros(self.n) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        Cf = np.zeros(18 * self.fine_n); T = np.zeros(self.fine_n)
        jc = C.c_double(); fo = C.c_double(); ne = C.c_long(); ni = C.c_int()
        ret = lib().orc_optimize(self.h, rel_cost_tol, max_iterations, x, int(x0 is not None), Cf, T, C.byref(jc), C.byref(fo),
                                 C.byref(ne), C.byref(ni))
        return dict(status=ret, x=x, C=Cf.reshape(-1, 3), T=T, jerk_cost=jc.value, objective=fo.value, evals=ne.value, iters=ni.value)


# ---------------------------------------------------------------------------------------------------------------------
# corridor generation: the reference's decomp_util compiled as it is, and the corridor loop restated
# ---------------------------------------------------------------------------------------------------------------------
_ref_decomp = None


def ref_decomp():
    """oracle/_ref/libref_decomp.so = the reference's decomp_util (line_segment.h, decomp_base.h, ellipsoid.h, polyhedron.h,
    ellipsoid_decomp.h) compiled where it lies against oracle/eigen_shim; None if absent."""
    global _ref_decomp
    if _ref_decomp is None:
        path = os.path.join(HERE, "_ref", "libref_decomp.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_line_segment_dilate.argtypes = [_dp, _dp, _dp, C.c_int, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int), _dp, _dp, _dp]
        R.ref_decomp_dilate.argtypes = [_dp, _dp, _dp, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), _dp]
        R.ref_poly_inside.argtypes = [C.c_int, _dp, _dp]
        _ref_decomp = R
    return _ref_decomp


def ref_line_segment_dilate(p1, p2, bbox, obs, offset=0.0, cap=4096):
    """LineSegment3D::dilate of the reference: (H 6 x K, ellipsoid C, d)."""
    R = ref_decomp()
    obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
    n = C.c_int(); rec = np.zeros(6 * cap); Cm = np.zeros(9); d = np.zeros(3)
    rc = R.ref_line_segment_dilate(np.ascontiguousarray(p1, dtype=np.float64), np.ascontiguousarray(p2, dtype=np.float64), np.ascontiguousarray(bbox, dtype=np.float64),
                                   len(obs), obs.ctypes.data if len(obs) else None, offset, cap, C.byref(n), rec, Cm, d)
    assert rc == 0
    return rec[:6 * n.value].reshape(-1, 6).T.copy(), Cm.reshape(3, 3), d


def corridor_oracle(path, obs, bbox, map_height, max_seg=4.0, blocked=None, dilate=None):
    """Restatement of the corridor loop of MavGlobalPlanner::plan (MinCoPlan_CPU.cpp:37-105) around a `dilate(p1, p2) -> H`
    callable (by default the reference's own EllipsoidDecomp3D through libref_decomp.so): list of 6 x K_i arrays."""
    path = np.asarray(path, dtype=np.float64).reshape(-1, 3)
    if dilate is None:
        dilate = lambda a, b: ref_line_segment_dilate(a, b, bbox, obs)[0]
    polys, n, i = [], len(path), 0
    while i < n - 1:                                                     # :44
        k = i + 1
        while k < n:                                                     # :47-52
            if (blocked is not None and blocked(path[i], path[k])) or np.linalg.norm(path[i] - path[k]) >= max_seg:
                k -= 1
                break
            k += 1
        if k < i + 1: k = i + 1                                          # :53-55
        if k >= n: k = n - 1                                             # :56
        H = dilate(path[i], path[k])                                     # :57-62
        j = k
        while j < n:                                                     # :67-75  Polyhedron::inside, 1e-10 slack
            sd = np.einsum("dk,dk->k", H[:3], path[j][:, None] - H[3:])
            if np.any(sd > 1e-10): break
            j += 1
        j -= 1
        zp = np.array([[0.0, 0.0, 1.0, 0.0, 0.0, map_height], [0.0, 0.0, -1.0, 0.0, 0.0, 0.0]]).T     # :85-91
        polys.append(np.concatenate([H, zp], axis=1))
        if j >= n - 1: break                                             # :77-79
        wp = (1 * i + 4 * j) // 5                                        # :81-82 (integer arithmetic)
        i = wp if wp > i else i + 1                                      # the reference does not advance when wp == i
    return polys


# ---------------------------------------------------------------------------------------------------------------------
# result type: the reference's Piece / Trajectory / RootFinder compiled as they are
# ---------------------------------------------------------------------------------------------------------------------
_ref_traj = None


def ref_traj():
    """oracle/_ref/libref_traj.so = trajectory.hpp + root_finder.hpp of the reference against oracle/eigen_shim; None if absent."""
    global _ref_traj
    if _ref_traj is None:
        path = os.path.join(HERE, "_ref", "libref_traj.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_piece_max_rates.argtypes = [C.c_double, _dp, _dp]
        R.ref_traj_eval.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp]
        R.ref_traj_max_rates.argtypes = [C.c_int, _dp, _dp, _dp]
        R.ref_piece_normalized.argtypes = [C.c_double, _dp, _dp]
        _ref_traj = R
    return _ref_traj


def piece_layout(Cf):
    """(6N x 3) coefficient rows (power k of piece i in row 6i+k) -> the reference Piece layout [piece][axis][column], column j = power 5-j."""
    pc = np.asarray(Cf, dtype=np.float64).reshape(-1, 6, 3)
    return np.ascontiguousarray(pc[:, ::-1, :].transpose(0, 2, 1))
