"""CPU-side algebra check of the product's per-sample math header (frx_math.hpp): the reverse-mode
adjoints must reproduce the reference's explicit-Jacobian penalty (via the oracle).  The header is
compiled for the host by tests/hostcheck only for this purpose; the product has no CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _load_hostcheck():
    d = os.path.join(ROOT, "tests", "hostcheck")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    H = C.CDLL(os.path.join(d, "libhostcheck.so"))
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"); ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    H.hostcheck_penalty.argtypes = [C.c_int, C.c_int, dp, dp, ip, dp, dp, C.c_int, dp]
    return H


@pytest.fixture(scope="module")
def hc():
    return _load_hostcheck()


@pytest.mark.parametrize("obst,kappa", [(False, 8), (True, 16)])
def test_reverse_mode_penalty_matches_oracle(hc, sc, ob, obst, kappa):
    P = sc.ZHANGJIAJIE
    c = sc.make_candidate(3, 24, 6, obstacles=obst)
    o = ob.Oracle(c, P, qd_intervals=kappa)
    h_off, h_rec, _, _ = c.packed()
    hr = h_rec.reshape(-1, 6).copy(); hr[:, :3] /= np.linalg.norm(hr[:, :3], axis=1)[:, None]
    pcv = np.array([P["horiz_half_len"], P["horiz_half_len"], P["vert_half_len"], P["safe_margin"], P["vel_max"], P["thr_acc_min"],
                    P["thr_acc_max"], P["body_rate_max"], P["grav_acc"], *P["penalty_pvtb"]])
    x0 = o.initial_guess()
    pts = [x0] + [o.optimize(1e-6, max_iterations=k, x0=x0)["x"] for k in (20, 80, 400)]
    for acc in (1, 0):
        o.set_abscissa_mode(bool(acc))
        for x in pts:
            T, _, Cf = o.forward(x)
            cost, gdT, gdC = o.penalty(T, Cf)
            out = np.zeros(20 * o.fine_n)
            hc.hostcheck_penalty(o.fine_n, kappa, T, np.ascontiguousarray(Cf.reshape(-1)), h_off, np.ascontiguousarray(hr.reshape(-1)), pcv, acc, out)
            out = out.reshape(-1, 20)
            assert abs(out[:, 0].sum() - cost) <= 1e-11 * max(abs(cost), 1e-300)
            assert np.abs(out[:, 1] - gdT).max() <= 1e-10 * max(np.abs(gdT).max(), 1e-300)
            assert np.abs(out[:, 2:].reshape(-1, 3) - gdC).max() <= 1e-10 * max(np.abs(gdC).max(), 1e-300)


# ---- host side of setup() + the reference's initial guess (frx_host_setup.hpp), the code frx_initial_guess runs, on a CPU-only box ----
def _host_guess(hc, frx, cands, params, kappa, threads=None):
    cfg = frx.FrxConfig.from_params(params, qd_intervals=kappa)
    coarse_n, ini, fin, h_off, h_rec, v_off, v_rec = frx.pack_batch(cands)
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"); ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    hc.hostcheck_initial_guess.restype = C.c_int
    hc.hostcheck_initial_guess.argtypes = [C.POINTER(frx.FrxConfig), C.c_int, ip, dp, dp, ip, dp, ip, dp, C.c_int]
    x_off = np.zeros(len(cands) + 1, np.int32); x0 = np.zeros(4096 * len(cands))
    old = os.environ.get("FRX_SETUP_THREADS")
    if threads is not None: os.environ["FRX_SETUP_THREADS"] = str(threads)
    try:
        n = hc.hostcheck_initial_guess(C.byref(cfg), len(cands), coarse_n, ini, fin, v_off, v_rec, x_off, x0, x0.size)
    finally:
        if threads is not None:
            if old is None: del os.environ["FRX_SETUP_THREADS"]
            else: os.environ["FRX_SETUP_THREADS"] = old
    assert n > 0
    return x_off, x0[:n].copy()


def test_initial_guess_of_the_host_code_matches_the_oracle_whatever_the_thread_count(hc, frx, sc, ob):
    """setInitial / backwardT / backwardP (CPU.hpp:1188-1228, 679-726, 777-813) as the library runs them - a flat task list over (candidate,
    waypoint) - against the oracle's restatement (itself pinned to the compiled reference at 1e-13, tests/test_reference_pin.py)."""
    P = sc.ZHANGJIAJIE
    cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(6)] + [sc.make_candidate(7, 12, 3, obstacles=True), sc.make_candidate(9, 1, 0)]
    x_off, x_ser = _host_guess(hc, frx, cands, P, 16, threads=1)
    for b, c in enumerate(cands):
        o = ob.Oracle(c, P, qd_intervals=16)
        ref = o.initial_guess()
        assert x_off[b + 1] - x_off[b] == ref.size
        assert np.abs(x_ser[x_off[b]:x_off[b + 1]] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    for nt in (2, 5, 8):                                           # the same solves in the same arithmetic: bit-identical for every partition
        _, x_par = _host_guess(hc, frx, cands, P, 16, threads=nt)
        assert np.array_equal(x_par, x_ser)
