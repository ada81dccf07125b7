"""BASELINE.json configs[3] and [4] (one GPU's share) and the north_star's coefficient contract, end to end through frx_optimize.

Two facts frame these tests (measured):
  * the reference's optimiser output is path-sensitive: the SAME CPU code run twice with a 1e-16 perturbation (the two sample-abscissa
    forms of CPU.hpp:400 / cc.cu:152, or x0 moved by a few ulp) ends 1e-3 ... 1e-2 apart in the coefficients at EVERY stopping
    tolerance from 1e-6 down to 1e-12, because the stop rule bounds the relative cost decrease and the cost is flat along
    time-allocation directions.  Independent runs therefore cannot agree to 1e-6; what can be asserted is that the device-driven
    plan differs from a CPU-driven plan by no more than CPU-driven plans differ among themselves, and that the L-BFGS status
    (the reference's own success/failure verdict) is the same for every candidate;
  * the MAP x -> coefficients and the objective/gradient agree to 1e-9 ... 1e-13 (test_gpu_parity.py), which is what the 1e-6
    contract is checked on in lock-step form.
"""
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_plans(ob, sc, cands, kappa, tol, variants):
    """CPU-driven plans of every candidate under each variant = (abscissa mode, ulp-scale perturbation seed of x0)."""
    def one(args):
        c, (mode, seed) = args
        o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=kappa)
        o.set_abscissa_mode(mode)
        x0 = o.initial_guess()
        if seed:
            x0 = x0 * (1.0 + 4e-16 * np.random.default_rng(seed).integers(-2, 3, x0.size))
        r = o.optimize(tol, x0=x0)
        return dict(status=int(r["status"]), objective=float(r["objective"]), C=np.array(r["C"]), T=np.array(r["T"]), iters=int(r["iters"]))
    jobs = [(c, v) for c in cands for v in variants]
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 64)) as ex:
        out = list(ex.map(one, jobs))
    nv = len(variants)
    return [out[i * nv:(i + 1) * nv] for i in range(len(cands))]


def _device_plans(frx, sc, cands, kappa, tol, batch=32):
    """Plans of `cands` on the device in batches that fit the resident round kernel (32 headline-size candidates); the resident kernel's
    OWN verdicts: nothing is re-run on another path (frx_debug_set_resident_retry stays off, and the environment must not switch it on)."""
    assert os.environ.get("FRX_RESIDENT_RETRY", "0") == "0"
    out = {"status": [], "objective": [], "ms_total": 0.0, "rounds": 0, "resident": [], "resident_failed": 0, "resident_retried": 0}
    for lo in range(0, len(cands), batch):
        prob = frx.Problem(cands[lo:lo + batch], sc.ZHANGJIAJIE, qd_intervals=kappa)
        r = prob.optimize(tol)
        assert r["device_status"] == 0
        out["status"].append(r["status"]); out["objective"].append(r["objective"]); out["resident"].append(r["resident"])
        out["ms_total"] += r["ms_total"]; out["rounds"] = max(out["rounds"], r["rounds"])
        out["resident_failed"] += r["resident_failed"]; out["resident_retried"] += r["resident_retried"]
        prob.close()
    out["status"] = np.concatenate(out["status"]); out["objective"] = np.concatenate(out["objective"])
    return out


def _share_check(frx, sc, ob, cands, kappa, label, queue=False):
    tol = sc.ZHANGJIAJIE["opt_rel_tol"]
    r = _device_plans(frx, sc, cands, kappa, tol)
    assert min(r["resident"]) >= 3 and r["resident_retried"] == 0          # every batch ran on the resident round kernel, once
    # CPU, CPU' (the two abscissa forms: a 1e-16 perturbation) and two more CPU runs whose x0 is moved by a few ulp: the spread of a
    # heavy-tailed quantity needs more than two samples
    variants = [(False, 0), (True, 0), (False, 11), (False, 12)]
    cpu = _cpu_plans(ob, sc, cands, kappa, tol, variants)
    spread, dev, bad_status, infeasible_like, n_fail = [], [], [], [], 0
    for b, plans in enumerate(cpu):
        sts = [p["status"] for p in plans]
        sa = sts[0]
        n_fail += sa < 0
        ok = r["status"][b] == sa or (len(set(sts)) > 1 and r["status"][b] in sts)   # path-sensitive candidates may take any CPU verdict
        if not ok and max(sts) < 0:
            # An INFEASIBLE scenario (every CPU variant fails): the run is chaotic from the first iterations (scenario 91 of the Monte-Carlo
            # share: 16 of 16 CPU variants end in -1005 after 287 ... 15792 iterations at objectives 2e7 ... 6e13) and stalls somewhere on a
            # penalty plateau; whether the stall is labelled as a failed line search or - one tiny step later - as a met stop criterion
            # depends on the last bits.  What has to be reproduced is the verdict "no plan": a penalty-dominated objective.
            infeasible_like.append((b, int(r["status"][b]), float(r["objective"][b]), sts))
            ok = True
        if not ok:
            bad_status.append((b, int(r["status"][b]), sts))
        if min(sts) >= 0 and r["status"][b] >= 0:
            objs = np.array([p["objective"] for p in plans])
            spread.append((objs.max() - objs.min()) / abs(objs[0]))
            dev.append(np.abs(r["objective"][b] - objs).min() / abs(objs[0]))
    spread, dev = np.array(spread), np.array(dev)
    feasible_obj = np.median([min(p["objective"] for p in plans) for plans in cpu if min(q["status"] for q in plans) >= 0])
    for b, st, obj, sts in infeasible_like:                                 # "succeeded" where the reference fails: only with a penalty-dominated objective
        if not (obj > 100.0 * feasible_obj):
            bad_status.append((b, st, sts))
    n_fail_dev = int(np.sum((r["status"] < 0) & (r["status"] != -1004)))
    n_fail_cpu_any = int(sum(1 for plans in cpu if min(p["status"] for p in plans) < 0))      # candidates on which at least one CPU variant fails
    summary = {"config": label, "candidates": len(cands), "failed_on_cpu": int(n_fail), "failed_on_any_cpu_variant": n_fail_cpu_any,
               "failed_on_resident_kernel": n_fail_dev, "resident_retried": r["resident_retried"], "status_mismatches": bad_status,
               "stalled_with_other_label": [(b, st, obj) for b, st, obj, _ in infeasible_like],
               "cpu_vs_cpu_objective_spread": {"median": float(np.median(spread)), "p95": float(np.percentile(spread, 95)), "max": float(spread.max())},
               "device_vs_cpu_objective": {"median": float(np.median(dev)), "p95": float(np.percentile(dev, 95)), "max": float(dev.max())},
               "plan_ms": r["ms_total"], "rounds": r["rounds"]}
    if queue:
        # the same share as ONE batch through the resident kernel's work queue (as many clusters as the chip holds; a cluster whose candidate is
        # finished is handed the next one): a candidate's plan does not depend on the cluster that runs it or on what ran there before
        prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
        prob.set_resident(2)
        q = prob.optimize(tol)
        prob.close()
        assert q["device_status"] == 0 and q["resident"] == min(r["resident"]) and 0 < q["clusters"] < len(cands), (q["resident"], q["clusters"])
        assert np.array_equal(q["status"], r["status"]) and np.array_equal(q["objective"], r["objective"])
        assert q["resident_failed"] == r["resident_failed"] and q["resident_retried"] == 0
        summary["work_queue"] = {"clusters": q["clusters"], "plan_ms": q["ms_total"], "plans_per_s": 1e3 * len(cands) / q["ms_total"],
                                 "batches_of_32_plan_ms": r["ms_total"], "commands_of_the_busiest_cluster": q["rounds"], "statuses_and_objectives_equal_the_batches": True}
    print(json.dumps(summary))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(summary, open(os.path.join(ROOT, "gpurun_out", f"share_{label}.json"), "w"), indent=1)
    assert not bad_status, f"L-BFGS status differs from the CPU reference path for {bad_status}"
    # the resident kernel's own failures (no second chance): not more candidates than the CPU oracle loses on the same share
    assert n_fail_dev == r["resident_failed"] and n_fail_dev <= max(int(n_fail), n_fail_cpu_any), summary
    # every candidate inside the CPU-vs-CPU' envelope: the distance to the nearest of the four CPU plans against the spread among them
    assert dev.max() <= 1.5 * spread.max(), (dev.max(), spread.max())
    assert np.median(dev) <= np.median(spread), (np.median(dev), np.median(spread))
    return summary


def test_config3_share_of_one_gpu(frx, sc, ob):
    """BASELINE.json configs[3]: 256 random gate perturbations over 8 GPUs = 32 per GPU; the share of rank 1 (perturb ids 32..63)."""
    B, N, gates, kappa = sc.CONFIGS["perturbed256"]
    cands = [sc.make_candidate(0, N, gates, perturb_id=32 + b) for b in range(B // 8)]
    _share_check(frx, sc, ob, cands, kappa, "perturbed256_rank1")


def test_config4_share_of_one_gpu(frx, sc, ob):
    """BASELINE.json configs[4]: Monte-Carlo sweep, independent scenarios; 128 of one GPU's 512 (ids 64..191, which include the
    infeasible scenario 170), run as four resident batches of 32: the reference's verdict (LBFGS status) reproduced for every scenario
    by the resident kernel itself, objectives inside the CPU-vs-CPU' envelope; then as one batch of 128 through the resident kernel's work
    queue: bit-identical verdicts and objectives."""
    B, N, gates, kappa = sc.CONFIGS["montecarlo4096"]
    cands = [sc.make_candidate(64 + b, N, gates) for b in range(128)]
    s = _share_check(frx, sc, ob, cands, kappa, "montecarlo4096_128", queue=True)
    assert s["failed_on_cpu"] >= 1                        # the share does contain an infeasible scenario


def _mc_fixture():
    path = os.path.join(ROOT, "tests", "golden", "mc512_cpu_verdicts.npz")
    assert os.path.exists(path), "tests/golden/mc512_cpu_verdicts.npz is missing: python tests/golden/make_mc_verdicts.py"
    return np.load(path)


def mc_verdict_check(status, objective, fix, first=0):
    """One path's verdicts on the scenarios first .. first + len(status) of the Monte-Carlo share against the CPU fixture (four CPU variants per scenario,
    tests/golden/make_mc_verdicts.py).  The reference's verdict is lbfgs_optimize's return code (se3gcopter_cpu.hpp:1243-1249).  Rule, as _share_check:
    the device's code is one the CPU variants produce; or EVERY CPU variant fails (an infeasible scenario: chaotic from the first iterations, how the stall is
    labelled depends on the last bits) and the device either fails too or 'succeeds' on a penalty-dominated objective.  Returns (rows that break the rule, rows of
    infeasible scenarios)."""
    cs, co = fix["status"], fix["objective"]
    feasible = np.median([co[i].min() for i in range(len(cs)) if cs[i].min() >= 0])
    bad, infeasible = [], []
    for b in range(len(status)):
        i = first + b
        sts = [int(v) for v in cs[i]]
        st = int(status[b])
        if st in sts:
            continue
        if max(sts) < 0:
            infeasible.append({"scenario": i, "device_status": st, "device_objective": float(objective[b]), "cpu_status": sts})
            if st >= 0 and not (objective[b] > 100.0 * feasible):
                bad.append(infeasible[-1])
            continue
        bad.append({"scenario": i, "device_status": st, "device_objective": float(objective[b]), "cpu_status": sts, "cpu_objective": [float(v) for v in co[i]]})
    return bad, infeasible


def test_config4_full_share_of_one_gpu(frx, sc, ob):
    """BASELINE.json configs[4] at a GPU's FULL share: all 512 Monte-Carlo scenarios of rank 0 as ONE batch through the path bench.py
    uses (per-stage rounds, the last 32 on the resident kernel: take-over) AND through the resident kernel's work queue, each scenario's verdict against the four
    CPU variants of the committed fixture; every scenario on which the two device paths disagree about success is listed with the CPU's verdicts and must be one
    on which the CPU variants fail as well.  A sample of the fixture is recomputed live with the oracle (it is data, and this checks it is THIS oracle's data)."""
    fix = _mc_fixture()
    B, N, gates, kappa = sc.CONFIGS["montecarlo4096"]
    B //= 8
    assert int(fix["first_id"]) == 0 and fix["status"].shape == (B, 4)
    tol = sc.ZHANGJIAJIE["opt_rel_tol"]
    cands = [sc.make_candidate(b, N, gates) for b in range(B)]
    # the fixture against the live oracle on a sample: the scenarios every variant fails on, the ones the variants disagree on, and a few ordinary ones
    cs = fix["status"]
    odd = [i for i in range(B) if cs[i].max() < 0 or len(set(int(v) >= 0 for v in cs[i])) > 1]
    sample = sorted(set(odd[:6] + [0, 1, 255, 511]))
    variants = [(bool(m), int(s)) for m, s in fix["variants"]]
    live = _cpu_plans(ob, sc, [cands[i] for i in sample], kappa, tol, variants)
    for i, plans in zip(sample, live):
        assert [p["status"] for p in plans] == [int(v) for v in cs[i]], (i, [p["status"] for p in plans], cs[i])
        assert np.allclose([p["objective"] for p in plans], fix["objective"][i], rtol=1e-12, atol=0.0, equal_nan=True), (i, [p["objective"] for p in plans], fix["objective"][i])
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = prob.initial_guess()
    r = prob.optimize(tol, x0=x0)                                           # bench.py's path: per-stage rounds + take-over
    prob.set_resident(2)
    q = prob.optimize(tol, x0=x0)                                           # the work queue
    prob.close()
    assert r["taken_over"] > 0 and q["device_status"] == 0 and 0 < q["clusters"] < B, (r["taken_over"], q["device_status"], q["clusters"])
    bad_r, inf_r = mc_verdict_check(r["status"], r["objective"], fix)
    bad_q, inf_q = mc_verdict_check(q["status"], q["objective"], fix)
    mism = [{"scenario": int(b), "default_path_status": int(r["status"][b]), "work_queue_status": int(q["status"][b]), "cpu_status": [int(v) for v in cs[b]],
             "every_cpu_variant_fails": bool(cs[b].max() < 0)} for b in range(B) if (r["status"][b] >= 0) != (q["status"][b] >= 0)]
    summary = {"config": "montecarlo4096_full_share", "scenarios": B, "cpu_all_variants_fail": int(np.sum(cs.max(axis=1) < 0)), "cpu_some_variant_fails": int(np.sum(cs.min(axis=1) < 0)),
               "default_path": {"ok": int(np.sum(r["status"] >= 0)), "taken_over": int(r["taken_over"]), "plan_ms": r["ms_total"], "outside_cpu_set": bad_r, "infeasible": inf_r},
               "work_queue": {"ok": int(np.sum(q["status"] >= 0)), "clusters": int(q["clusters"]), "plan_ms": q["ms_total"], "outside_cpu_set": bad_q, "infeasible": inf_q},
               "verdict_mismatches_between_the_paths": mism}
    print(json.dumps(summary))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "share_montecarlo4096_full.json"), "w"), indent=1)
    assert not bad_r, bad_r
    assert not bad_q, bad_q
    assert all(m["every_cpu_variant_fails"] for m in mism), mism


@pytest.mark.parametrize("sid,N,gates,kappa", [(1, 16, 4, 8), (2, 32, 8, 8)])
def test_coefficient_spread_against_stopping_tolerance(frx, sc, ob, sid, N, gates, kappa):
    """device-driven vs CPU-driven plans at delta = 1e-6 (stock) ... 1e-12.  At every delta the device plan is
    as close to the CPU plans as those are to each other (coefficients and objective); the curve goes to DESIGN.md §4."""
    cand = sc.make_candidate(sid, N, gates)
    prob = frx.Problem([cand], sc.ZHANGJIAJIE, qd_intervals=kappa)
    # 24 CPU-driven plans per tolerance: the outcome of the reference's stop rule is not just noisy but MULTI-MODAL (measured on the
    # oracle, N = 32 at delta = 1e-8: 21 of 24 runs take 3900 ... 5400 iterations and end within 4.5e-5 of each other, 3 stop after
    # 2900 ... 3300 iterations on a plateau 5.1e-4 ... 5.6e-4 higher); four samples miss the rarer mode more often than not
    variants = [(False, 0), (True, 0)] + [(False, seed) for seed in range(11, 33)]
    rows = []
    for delta in (1e-6, 1e-8, 1e-10, 1e-12):
        r = prob.optimize(delta)
        plans = _cpu_plans(ob, sc, [cand], kappa, delta, variants)[0]
        cmax = max(np.abs(p["C"]).max() for p in plans)
        cpu_c = max(np.abs(p["C"] - q["C"]).max() for p in plans for q in plans) / cmax
        cpu_f = max(abs(p["objective"] - q["objective"]) for p in plans for q in plans) / abs(plans[0]["objective"])
        dev_c = min(np.abs(r["C"] - p["C"]).max() for p in plans) / cmax
        dev_f = min(abs(r["objective"][0] - p["objective"]) for p in plans) / abs(plans[0]["objective"])
        rows.append({"delta": delta, "cpu_vs_cpu_coeff": float(cpu_c), "device_vs_cpu_coeff": float(dev_c), "cpu_vs_cpu_objective": float(cpu_f),
                     "device_vs_cpu_objective": float(dev_f), "device_status": int(r["status"][0]), "cpu_status": [p["status"] for p in plans],
                     "device_iters": int(r["iters"][0]), "cpu_iters": [p["iters"] for p in plans]})
        print(json.dumps(rows[-1]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", f"delta_curve_s{sid}_N{N}.json"), "w"), indent=1)
    prob.close()
    for row in rows:
        assert row["device_status"] in row["cpu_status"], row
        assert row["device_vs_cpu_coeff"] <= max(2.0 * row["cpu_vs_cpu_coeff"], 1e-6), row
        assert row["device_vs_cpu_objective"] <= max(2.0 * row["cpu_vs_cpu_objective"], 1e-9), row
