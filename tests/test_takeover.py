"""Take-over: the stragglers of a per-stage batch continue on the resident round kernel with their history as it stands.
CPU: the host code that rebuilds the resident kernel's dense state (R^-1 by slot, Y^T Y, D) from history rows, against dense linear algebra, and the
direction it implies against the two-loop recursion.  GPU: a plan that changes paths mid-way against the same plan on the per-stage rounds alone."""
import ctypes as C
import json
import os

import numpy as np
import pytest


def _compact(frx, m, n, hs, bound, newest, S, Y):
    rinv = np.zeros(128 * 129); yy = np.zeros(128 * 128); vd = np.zeros(128)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = frx.lib().frx_debug_compact_from_history(m, n, hs, bound, newest, fp(S), fp(Y), fp(rinv), fp(yy), fp(vd))
    assert rc == 0
    return rinv.reshape(128, 129)[:, :128], yy.reshape(128, 128), vd


def _two_loop(S, Y, g, newest, bound, m):
    d = -g.copy(); alpha = np.zeros(m); j = (newest + 1) % m
    for _ in range(bound):
        j = (j + m - 1) % m
        alpha[j] = (S[j] @ d) / (Y[j] @ S[j]); d -= alpha[j] * Y[j]
    d *= (Y[newest] @ S[newest]) / (Y[newest] @ Y[newest])
    for _ in range(bound):
        beta = (Y[j] @ d) / (Y[j] @ S[j]); d += (alpha[j] - beta) * S[j]; j = (j + 1) % m
    return d


@pytest.mark.parametrize("bound,newest", [(128, 37), (128, 127), (50, 49), (1, 0), (77, 5)])
def test_dense_state_rebuilt_from_a_history(frx, bound, newest):
    m, n, hs = 128, 641, 768
    rng = np.random.default_rng(bound * 131 + newest)
    S = np.zeros((m, hs)); Y = np.zeros((m, hs))
    A = rng.normal(size=(n, n)); H = A @ A.T / n + np.eye(n)                      # y = H s: s.y > 0 like on a convex stretch
    slots = [(newest - a) % m for a in range(bound)]                              # newest first
    S[:, :n] = 7.0; Y[:, :n] = -3.0                                               # stale rows of an earlier plan: must be ignored
    for j in slots:
        S[j, :n] = rng.normal(size=n); Y[j, :n] = H @ S[j, :n]
    rinv, yy, vd = _compact(frx, m, n, hs, bound, newest, S, Y)
    order = slots[::-1]                                                           # oldest ... newest
    R = np.triu(S[order] @ Y[order].T)
    Ri = np.linalg.inv(R)
    got = rinv[np.ix_(order, order)]
    assert np.abs(got - Ri).max() <= 1e-10 * np.abs(Ri).max()
    other = np.ones((128, 128), bool); other[np.ix_(order, order)] = False
    assert not rinv[other].any() and not yy[other].any()                           # slots without a pair: zeros (the kernel multiplies, it does not select)
    assert np.abs(yy[np.ix_(order, order)] - Y[order] @ Y[order].T).max() <= 1e-12 * np.abs(yy).max()
    assert np.allclose(vd[order], np.einsum("ij,ij->i", S[order], Y[order]), rtol=1e-13) and not vd[[j for j in range(128) if j not in order]].any()
    # the direction of the compact form with this state (rk_dense_loop: w = R^-1 S^T g, v = (D + gamma Y^T Y) w - gamma Y^T g, u = R^-T v, d = -gamma g - S u + gamma Y w)
    g = rng.normal(size=n)
    So, Yo = S[order][:, :n], Y[order][:, :n]
    gamma = (Yo[-1] @ So[-1]) / (Yo[-1] @ Yo[-1])
    w = got @ (So @ g)
    v = (np.diag(vd[order]) + gamma * yy[np.ix_(order, order)]) @ w - gamma * (Yo @ g)
    u = got.T @ v
    d = -gamma * g - So.T @ u + gamma * (Yo.T @ w)
    d_ref = _two_loop(S[:, :n], Y[:, :n], g, newest, bound, m)
    assert np.abs(d - d_ref).max() <= 1e-9 * np.abs(d_ref).max()


def _plan(prob, tol, env, takeover_at=0, **kw):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env); os.environ["FRX_TRACE"] = "1"
    prob.set_takeover_at(takeover_at)                       # frx_debug_set_takeover_at: per-stage rounds first, hand-over after so many (0: the library's own rule)
    try:
        r = prob.optimize(tol, **kw)
    finally:
        prob.set_takeover_at(0)
        os.environ.pop("FRX_TRACE", None)
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    r["trace"] = prob.trace()
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("B,at", [(1, 150), (3, 40), (20, 400)])
def test_a_plan_that_changes_paths_midway_follows_the_per_stage_plan(frx, sc, B, at):
    """frx_debug_set_takeover_at(p, k) hands a small batch over after k per-stage rounds (k = 400: the history has wrapped around three times; 150: once; 40: it is not full).
    From there the resident kernel runs on the SAME pairs: command by command the scalars that cross the mailbox - step, f, x.x, g.g and the new direction's
    slope g_p.d - equal the pure per-stage plan's to rounding for the next commands (they drift apart later like any two runs of the reference's stop rule),
    and the complete plans end with the same verdicts."""
    cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
    x0 = prob.initial_guess()
    tol = sc.ZHANGJIAJIE["opt_rel_tol"]
    off = {"FRX_RESIDENT": "0"}
    a = _plan(prob, tol, {}, takeover_at=at, x0=x0, max_iterations=at + 80)
    b = _plan(prob, tol, off, x0=x0, max_iterations=at + 80)
    assert b["resident"] == 0 and b["taken_over"] == 0
    assert a["taken_over"] == B and a["resident"] > 0 and a["device_status"] == 0, (a["taken_over"], a["resident"], a["device_status"])
    ta, tb = a["trace"], b["trace"]
    rows = min(len(ta), len(tb))
    assert rows >= at + 25, (len(ta), len(tb))
    worst_before = worst_after = 0.0
    series = []
    for i in range(rows):
        fa, fb = ta[i], tb[i]
        if i >= at + 30: break
        assert int(fa[0]) == int(fb[0]), f"command {i}: flags {fa[0]} vs {fb[0]}"
        errs = [abs(fa[1] - fb[1]) / max(abs(fb[1]), 1e-300), abs(fa[2] - fb[2]) / abs(fb[2]), abs(fa[5] - fb[5]) / max(fb[5], 1e-300), abs(fa[6] - fb[6]) / max(fb[6], 1e-300)]
        if int(fb[0]) & 4: errs.append(abs(fa[4] - fb[4]) / max(abs(fb[4]), 1e-300))
        if i < at: worst_before = max(worst_before, max(errs))
        else: worst_after = max(worst_after, max(errs)); series.append(float(max(errs)))
    print(json.dumps({"B": B, "hand_over_after_rounds": at, "worst_rel_diff_before": worst_before, "worst_rel_diff_in_the_30_commands_after": worst_after,
                      "per_command_after": [float(f"{v:.1e}") for v in series]}))
    assert worst_before == 0.0                                                  # the same per-stage rounds up to the hand-over
    # The first commands after the hand-over measure the hand-over itself (same pairs, the direction in the compact form instead of the two-loop recursion);
    # later ones the optimisation's own sensitivity - mid-plan a 1e-13 difference in a direction grows by one to two orders of magnitude per ten commands
    # (at the start of a plan it does not: tests/test_gpu_resident.py compares 30 commands at 1e-8).
    assert max(series[:4]) < 1e-9, series[:8]
    assert worst_after < 1e-2, worst_after
    full_a = _plan(prob, tol, {}, takeover_at=at, x0=x0)
    full_b = _plan(prob, tol, off, x0=x0)
    assert full_a["taken_over"] == B and np.array_equal(full_a["status"] >= 0, full_b["status"] >= 0) and np.all(full_a["status"] >= 0)
    rel = np.abs(full_a["objective"] - full_b["objective"]) / np.abs(full_b["objective"])
    assert rel.max() < 5e-3, rel
    prob.close()


@pytest.mark.gpu
def test_stragglers_of_a_large_batch_finish_on_the_resident_kernel(frx, sc):
    """110 headline-size candidates: more than three times the chip's 32 clusters, so the plan starts as per-stage rounds; when 32 are left they move to the
    resident kernel.  Same verdicts as the per-stage rounds alone, objectives as close as two runs of the stop rule are, and no slower."""
    B = 110
    cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
    x0 = prob.initial_guess()
    tol = sc.ZHANGJIAJIE["opt_rel_tol"]
    a = _plan(prob, tol, {}, x0=x0)
    b = _plan(prob, tol, {"FRX_TAKEOVER": "0"}, x0=x0)
    assert b["taken_over"] == 0 and b["resident"] == 0
    assert 0 < a["taken_over"] <= 32 and a["device_status"] == 0, (a["taken_over"], a["device_status"])
    assert np.array_equal(a["status"] >= 0, b["status"] >= 0)
    rel = np.abs(a["objective"] - b["objective"]) / np.abs(b["objective"])
    print(json.dumps({"candidates": B, "taken_over": a["taken_over"], "plan_ms_with_take_over": a["ms_total"], "plan_ms_per_stage_only": b["ms_total"], "objective_rel_diff_max": float(rel.max()),
                      "evals_max": int(b["evals"].max()), "evals_mean": float(b["evals"].mean())}))
    assert rel.max() < 5e-3
    assert a["ms_total"] <= 1.05 * b["ms_total"]
    prob.close()
