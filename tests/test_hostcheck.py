"""CPU-side algebra check of the product's per-sample math header (frx_math.hpp): the reverse-mode
adjoints must reproduce the reference's explicit-Jacobian penalty (via the oracle).  The header is
compiled for the host by tests/hostcheck only for this purpose; the product has no CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def hc():
    d = os.path.join(ROOT, "tests", "hostcheck")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    H = C.CDLL(os.path.join(d, "libhostcheck.so"))
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"); ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    H.hostcheck_penalty.argtypes = [C.c_int, C.c_int, dp, dp, ip, dp, dp, C.c_int, dp]
    return H


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
