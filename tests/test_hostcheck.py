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


def test_history_row_stride_rule(hc):
    """k_lbfgs_pre's history rows (round 6): as long as the vector needs - n + 2 rounded up to 16 doubles, so that the row's last pair lies in its zero tail - when that is
    shorter than the workgroup's shape 64 W E, the shape has at least 512 doubles, and only a thread's LAST pair can fall beyond the end (the kernel clamps that one)."""
    hc.hostcheck_dv_rows.argtypes = [C.c_int, C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
    out = np.zeros(4, np.int32)
    for n in list(range(2, 2100, 7)) + [511, 512, 513, 639, 640, 641, 655, 656, 657, 767, 768, 769, 1024, 1025]:
        hc.hostcheck_dv_rows(n, 1, out)
        E, W, shape, stride = (int(v) for v in out)
        if E == 0: continue
        assert shape == 64 * W * E >= n
        assert stride <= shape and stride % 2 == 0 and stride >= min(shape, n + 2)
        if stride < shape:
            assert shape >= 512 and stride % 16 == 0 and stride - n < 18
            assert stride // 2 > 64 * W * (E // 2 - 1)                      # every slab of pairs but the last lies inside the row
        hc.hostcheck_dv_rows(n, 0, out)
        assert int(out[3]) == shape
    hc.hostcheck_dv_rows(641, 1, out)
    assert list(out) == [6, 2, 768, 656]                                   # the headline vector
    hc.hostcheck_dv_rows(639, 1, out)
    assert int(out[2]) == 640 == int(out[3])                               # the Monte-Carlo scenarios: already tight


def test_solo_launch_lds_layout(hc):
    """The solo launch's LDS (frx_solo_layout.hpp): resident operands in front, the polytopes LAST before the bodies' scratch so that the penalty phase's corridor blocks and
    transpose square can lie over both; 16-byte aligned regions; the headline / Monte-Carlo geometry fits two workgroups on a CU (80 KB each)."""
    hc.hostcheck_solo_lds.argtypes = [C.c_int] * 7 + [np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
    out = np.zeros(8, np.int32)
    for maxN, maxXb, maxVb, maxCN, nsteps, ppg, Kmax in [(64, 641, 2100, 16, 6, 15, 8), (64, 639, 2058, 16, 6, 15, 8), (64, 700, 2400, 16, 6, 5, 14), (12, 90, 300, 3, 4, 28, 8), (64, 641, 2100, 16, 6, 128, 8)]:
        hc.hostcheck_solo_lds(maxN, maxXb, maxVb, maxCN, nsteps, ppg, Kmax, out)
        ctl, xs, pw, wq, vs, ev, total, pen = (int(v) for v in out)
        assert 0 == ctl < xs < pw < wq < vs < ev < total and all(v % 2 == 0 for v in (xs, pw, wq, vs, ev, total))
        assert xs - ctl >= 19 * maxN and pw - xs >= maxXb and wq - pw >= (8 * nsteps + 5) * 64 and vs - wq >= 256 and ev - vs >= maxVb
        assert total - ev >= (24 * 64 + 4) + 9 * 65 + 2 * 64 + maxCN + 10    # the adjoint's scratch: rows | knot arrays | Tf, gT | gCo | partials
        assert total - vs >= pen                                               # the penalty phase's scratch lies over the polytopes and the bodies' scratch
    hc.hostcheck_solo_lds(64, 639, 2058, 16, 6, 15, 8, out)
    assert 8 * int(out[6]) <= 80 * 1024
