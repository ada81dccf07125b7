"""Three implementations of the same solver must produce IDENTICAL iterates on the oracle objective:
  (1) the reference's own lbfgs.hpp compiled where it lies (oracle/_ref/libref_lbfgs.so),
  (2) the oracle's blocking restatement (oracle/lbfgs_port.hpp),
  (3) the product's resumable batched state machine (fast-racing_amd/csrc/frx_lbfgs.hpp via frx_lbfgs_minimize_batch).
This pins (2) and (3) to the reference bit for bit (same compiler flags: -O3 -march=x86-64-v3)."""
import ctypes as C

import numpy as np
import pytest


def run_trace(fn, o, ob, x0, pm, cap=6000):
    L = ob.lib()
    x = x0.copy(); fx = C.c_double()
    tf = np.zeros(cap); ts = np.zeros(cap); tl = np.zeros(cap, np.int32); n = C.c_int()
    ret = fn(o.n, x, C.byref(fx), L.orc_objective_fnptr(), o.h, pm, cap, tf, ts, tl, C.byref(n))
    return ret, x, fx.value, tf[:n.value], ts[:n.value], tl[:n.value]


def product_minimize(frx, oracles, x0s, pm_struct, n_threads=2):
    x_off = np.zeros(len(oracles) + 1, np.int32)
    for i, o in enumerate(oracles):
        x_off[i + 1] = x_off[i] + o.n
    x = np.concatenate(x0s).astype(np.float64)
    calls = []

    def cb(inst, n_active, ids, xp, fp, gp):
        act = [ids[k] for k in range(n_active)]
        calls.append(len(act))
        for i in act:
            lo, hi = x_off[i], x_off[i + 1]
            xi = np.ctypeslib.as_array(xp, shape=(x_off[-1],))[lo:hi].copy()
            f, g = oracles[i].objective(xi)
            fp[i] = f
            np.ctypeslib.as_array(gp, shape=(x_off[-1],))[lo:hi] = g

    cbf = frx.BATCH_EVAL_FN(cb)
    n = len(oracles)
    f = np.zeros(n); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32); ev = np.zeros(n, np.int32)
    rc = frx.lib().frx_lbfgs_minimize_batch(n, x_off, x, f, st, it, ev, C.byref(pm_struct), cbf, None, n_threads)
    assert rc == 0
    return x, f, st, it, ev, x_off, calls


@pytest.fixture(scope="module")
def problems(sc, ob):
    cs = [sc.make_candidate(7, 12, 3, perturb_id=b, obstacles=(b == 1)) for b in range(3)]
    return [ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=8) for c in cs]


def test_port_is_bit_identical_to_reference_solver(problems, ob):
    R = ob.ref_lbfgs()
    if R is None:
        pytest.skip("oracle/_ref/libref_lbfgs.so not built (needs /root/reference at build time)")
    o = problems[0]
    x0 = o.initial_guess()
    pm = ob.lbfgs_params(mem_size=128, past=3, g_epsilon=1e-16, min_step=1e-32, delta=1e-6)
    a = run_trace(ob.lib().orc_lbfgs_run, o, ob, x0, pm)
    b = run_trace(R.ref_lbfgs_run, o, ob, x0, pm)
    assert a[0] == b[0] and a[2] == b[2]
    assert len(a[3]) == len(b[3]) > 50
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


@pytest.mark.parametrize("delta,mem", [(1e-6, 128), (1e-4, 8)])
def test_state_machine_is_bit_identical_to_blocking_solvers(frx, problems, ob, delta, mem):
    x0s = [o.initial_guess() for o in problems]
    pm = ob.lbfgs_params(mem_size=mem, past=3, g_epsilon=1e-16, min_step=1e-32, delta=delta)
    ps = frx.LbfgsParams()
    frx.lib().frx_lbfgs_default_params(C.byref(ps))
    ps.mem_size = mem; ps.past = 3; ps.g_epsilon = 1e-16; ps.min_step = 1e-32; ps.delta = delta
    x, f, st, it, ev, x_off, calls = product_minimize(frx, problems, x0s, ps)
    R = ob.ref_lbfgs()
    runner = R.ref_lbfgs_run if R is not None else ob.lib().orc_lbfgs_run
    for i, o in enumerate(problems):
        ret, xr, fr, tf, ts, tl = run_trace(runner, o, ob, x0s[i], pm)
        assert st[i] == ret
        assert f[i] == fr
        assert np.array_equal(x[x_off[i]:x_off[i + 1]], xr)
        assert 1 + tl.sum() <= ev[i] <= 1 + tl.sum() + 80    # initial + line-search evaluations (+ those of a failed More-Thuente search before its backtracking fallback)
    assert max(calls) == len(problems) and min(calls) >= 1   # candidates share evaluation rounds, drop out when done


def test_gcopter_params_match_reference(frx):
    p = frx.gcopter_lbfgs_params(1e-6)                    # CPU.hpp:1243-1247 on top of lbfgs.hpp:128-140
    assert (p.mem_size, p.past, p.g_epsilon, p.min_step, p.delta) == (128, 3, 1e-16, 1e-32, 1e-6)
    assert (p.max_iterations, p.max_linesearch, p.max_step, p.f_dec_coeff, p.s_curv_coeff, p.xtol) == (0, 40, 1e20, 1e-4, 0.9, 1e-16)


def test_invalid_parameters_are_reported_like_the_reference(frx):
    ps = frx.LbfgsParams()
    frx.lib().frx_lbfgs_default_params(C.byref(ps))
    ps.mem_size = 0
    x_off = np.array([0, 2], np.int32); x = np.zeros(2); f = np.zeros(1)
    st = np.zeros(1, np.int32); it = np.zeros(1, np.int32); ev = np.zeros(1, np.int32)
    cb = frx.BATCH_EVAL_FN(lambda *a: None)
    assert frx.lib().frx_lbfgs_minimize_batch(1, x_off, x, f, st, it, ev, C.byref(ps), cb, None, 1) == 0
    assert st[0] == -1020                                  # LBFGSERR_INVALID_MEMSIZE (lbfgs.hpp:149-206)


def test_rosenbrock_and_already_minimized(frx):
    ps = frx.LbfgsParams()
    frx.lib().frx_lbfgs_default_params(C.byref(ps))
    x_off = np.array([0, 2, 4], np.int32)
    x = np.array([-1.2, 1.0, 1.0, 1.0])

    def cb(inst, n_active, ids, xp, fp, gp):
        xa = np.ctypeslib.as_array(xp, shape=(4,)); ga = np.ctypeslib.as_array(gp, shape=(4,))
        for k in range(n_active):
            i = ids[k]; a, b = xa[2 * i], xa[2 * i + 1]
            fp[i] = (1 - a) ** 2 + 100 * (b - a * a) ** 2
            ga[2 * i] = -2 * (1 - a) - 400 * a * (b - a * a); ga[2 * i + 1] = 200 * (b - a * a)

    f = np.zeros(2); st = np.zeros(2, np.int32); it = np.zeros(2, np.int32); ev = np.zeros(2, np.int32)
    assert frx.lib().frx_lbfgs_minimize_batch(2, x_off, x, f, st, it, ev, C.byref(ps), frx.BATCH_EVAL_FN(cb), None, 1) == 0
    assert st[0] == 0 and np.allclose(x[:2], 1.0, atol=1e-4)
    assert st[1] == 2 and ev[1] == 1                       # LBFGS_ALREADY_MINIMIZED


def test_device_vector_protocol_is_bit_identical_on_host_emulation(problems, ob):
    """SolverDV keeps only scalars + decisions on the host and sends vector commands to the device.  Driven by a host
    emulation of those commands (tests/hostcheck, same sequential loops as Solver) it must reproduce the reference
    solver's iterates bit for bit: pins the command protocol without a GPU."""
    import os, subprocess
    from conftest import ROOT
    d = os.path.join(ROOT, "tests", "hostcheck")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    H = C.CDLL(os.path.join(d, "libhostcheck.so"))
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    H.hostcheck_lbfgs_dv.restype = C.c_int
    H.hostcheck_lbfgs_dv.argtypes = [C.c_int, dp, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    R = ob.ref_lbfgs()
    runner = R.ref_lbfgs_run if R is not None else ob.lib().orc_lbfgs_run
    for mem, delta, maxit in ((128, 1e-6, 0), (8, 1e-5, 0), (128, 1e-6, 37)):
        pm = ob.lbfgs_params(mem_size=mem, past=3, g_epsilon=1e-16, min_step=1e-32, delta=delta, max_iterations=maxit)
        for o in problems:
            x0 = o.initial_guess()
            ret, xr, fr, tf, ts, tl = run_trace(runner, o, ob, x0, pm)
            x = x0.copy(); fx = C.c_double(); it = C.c_int(); ev = C.c_int()
            rc = H.hostcheck_lbfgs_dv(o.n, x, C.byref(fx), ob.lib().orc_objective_fnptr(), o.h, pm, C.byref(it), C.byref(ev))
            assert rc == ret and fx.value == fr and np.array_equal(x, xr)


def test_first_trial_shortcut_equals_the_line_search_state_machine():
    """LineSearch::first_trial_accepted - what the leader of the resident round kernel asks before it runs the More-Thuente state machine -
    says "accepted" exactly when mt_begin + one mt_feed do (lbfgs.hpp:743-788, 829-850): random and adversarial inputs, including values
    ON the two thresholds, non-finite values, wrong-signed slopes and a search limit of one."""
    import os, subprocess
    from conftest import ROOT
    d = os.path.join(ROOT, "tests", "hostcheck")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    H = C.CDLL(os.path.join(d, "libhostcheck.so"))
    H.hostcheck_first_trial.restype = C.c_int
    H.hostcheck_first_trial.argtypes = [C.c_double] * 5 + [C.c_int] + [C.c_double] * 5 + [C.POINTER(C.c_int)] * 2
    rng = np.random.default_rng(5)
    ftol, gtol, lo, hi, xtol = 1e-4, 0.9, 1e-20, 1e20, 1e-16
    n_acc = 0
    cases = []
    for _ in range(20000):
        step = float(10.0 ** rng.uniform(-6, 1)) if rng.random() < 0.7 else 1.0
        f0 = float(rng.normal() * 10.0 ** rng.uniform(-2, 6))
        dgi = -float(10.0 ** rng.uniform(-8, 4))
        ftest = f0 + step * (ftol * dgi)
        f = [ftest, np.nextafter(ftest, np.inf), np.nextafter(ftest, -np.inf), f0 + 2.0 * step * dgi * rng.random(), f0 + abs(f0) * 1e-3][rng.integers(5)]
        thr = gtol * (-dgi)
        dg = [thr, -thr, np.nextafter(thr, np.inf), -np.nextafter(thr, np.inf), dgi * rng.random(), -dgi * rng.random() * 2][rng.integers(6)]
        cases.append((step, f0, dgi, float(f), float(dg), 40))
    cases += [(1.0, 1.0, 1e-3, 0.5, 0.0, 40), (0.0, 1.0, -1.0, 0.5, 0.0, 40), (-1.0, 1.0, -1.0, 0.5, 0.0, 40), (1.0, 1.0, -1.0, float("nan"), 0.0, 40),
              (1.0, 1.0, -1.0, float("inf"), 0.0, 40), (1.0, 1.0, -1.0, -float("inf"), 0.0, 40), (1.0, 1.0, -1.0, 0.5, float("nan"), 40),
              (1.0, 1.0, -1.0, 0.5, 0.0, 1), (1.0, 1.0, 0.0, 1.0, 0.0, 40), (1.0, 1.0, -0.0, 1.0, 0.0, 40)]
    for step, f0, dgi, f, dg, mls in cases:
        a = C.c_int(); b = C.c_int()
        same = H.hostcheck_first_trial(ftol, gtol, lo, hi, xtol, mls, step, f0, dgi, f, dg, C.byref(a), C.byref(b))
        assert same == 1, (step, f0, dgi, f, dg, mls, a.value, b.value)
        n_acc += a.value
    assert 2000 < n_acc < len(cases) - 2000                  # both verdicts are well represented


def test_value_after_a_failed_line_search_belongs_to_the_restored_point(frx):
    """When the line search gives up for good, x and g are reverted to the previous point (lbfgs.hpp:1287-1288); the objective
    that is reported with them must be the one AT that point (it ranks candidates), not the last rejected trial's."""
    ps = frx.LbfgsParams()
    frx.lib().frx_lbfgs_default_params(C.byref(ps))
    w = np.array([1.0, 10.0, 100.0])
    calls = {"n": 0}

    def cb(inst, n_active, ids, xp, fp, gp):
        # a smooth bowl until the first step has been accepted, then a cliff (+1e30, finite): Armijo can never hold again, the
        # search shrinks to min_step and gives up; the last rejected trial's value is the cliff
        calls["n"] += 1
        xa = np.ctypeslib.as_array(xp, shape=(3,)); ga = np.ctypeslib.as_array(gp, shape=(3,))
        if calls["n"] <= 2:
            fp[0] = float(np.sum(w * (xa - 1.0) ** 2)); ga[:] = 2.0 * w * (xa - 1.0)
        else:
            fp[0] = 1e30; ga[:] = 2.0 * w * (xa - 1.0)

    x_off = np.array([0, 3], np.int32); x = np.zeros(3); f = np.zeros(1)
    st = np.zeros(1, np.int32); it = np.zeros(1, np.int32); ev = np.zeros(1, np.int32)
    assert frx.lib().frx_lbfgs_minimize_batch(1, x_off, x, f, st, it, ev, C.byref(ps), frx.BATCH_EVAL_FN(cb), None, 1) == 0
    assert st[0] < 0 and calls["n"] > 2                                      # the search failed for good
    f_at_x = float(np.sum(w * (x - 1.0) ** 2))
    assert abs(f[0] - f_at_x) <= 1e-12 * max(f_at_x, 1.0), (f[0], f_at_x, x)
