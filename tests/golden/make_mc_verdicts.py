"""Generates tests/golden/mc512_cpu_verdicts.npz: the CPU oracle's verdicts for one GPU's FULL share of BASELINE.json configs[4]
(Monte-Carlo sweep: scenario ids 0 .. 511 = rank 0 of 8, 64 pieces x kappa 16, stock tolerance) under four variants of the SAME CPU code -
the two sample-abscissa forms (CPU.hpp:400 / cc.cu:152: a 1e-16 perturbation) and two start points moved by a few ulp.  The reference's
verdict on a scenario is lbfgs_optimize's return code (se3gcopter_cpu.hpp:1243-1249); its optimiser output is path-sensitive,
so the set of the four variants' verdicts is what a device plan is checked against (tests/test_gpu_configs.py: test_config4_full_share_of_one_gpu).
2048 CPU plans are ~25 minutes of one core each eight at a time: too long for the GPU box's test run, hence a committed fixture; the GPU test
re-runs a sample of the scenarios live against it.  Run from the repo root (oracle/liboracle.so built):
    python tests/golden/make_mc_verdicts.py [first_id] [count]
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

VARIANTS = [(False, 0), (True, 0), (False, 11), (False, 12)]       # (abscissa accumulated instead of multiplied, seed of the ulp-scale perturbation of x0): as _share_check
ITERATION_CAP = 60000                                              # tests/conftest.py: the reference runs unbounded and an infeasible scenario can loop for ever on NaN objectives (scenario 170); the cap shows up as -1004


def plans_of(sid):
    from frx_import import frx  # noqa: F401
    from fast_racing_amd import scenario as sc
    from oracle import binding as ob
    B, N, gates, kappa = sc.CONFIGS["montecarlo4096"]
    cand = sc.make_candidate(sid, N, gates)
    out = []
    for mode, seed in VARIANTS:
        o = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=kappa)
        o.set_abscissa_mode(mode)
        x0 = o.initial_guess()
        if seed:
            x0 = x0 * (1.0 + 4e-16 * np.random.default_rng(seed).integers(-2, 3, x0.size))
        r = o.optimize(sc.ZHANGJIAJIE["opt_rel_tol"], max_iterations=ITERATION_CAP, x0=x0)
        out.append((int(r["status"]), float(r["objective"]), int(r["iters"]), int(r["evals"])))
    return sid, out


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    ids = list(range(first, first + count))
    status = np.zeros((count, len(VARIANTS)), dtype=np.int32)
    objective = np.zeros((count, len(VARIANTS)))
    iters = np.zeros((count, len(VARIANTS)), dtype=np.int32)
    evals = np.zeros((count, len(VARIANTS)), dtype=np.int32)
    # resumable: finished scenarios are kept in a scratch file next to the fixture (deleted at the end)
    import json
    part_path = os.path.join(ROOT, "tests", "golden", ".mc512_partial.jsonl")
    have = {}
    if os.path.exists(part_path):
        for line in open(part_path):
            rec = json.loads(line)
            have[rec["sid"]] = rec["plans"]
    todo = [sid for sid in ids if sid not in have]
    with ProcessPoolExecutor(max_workers=int(os.environ.get("FRX_GOLDEN_WORKERS", os.cpu_count() or 1))) as ex, open(part_path, "a") as part:
        for n, (sid, plans) in enumerate(ex.map(plans_of, todo, chunksize=2)):
            have[sid] = plans
            part.write(json.dumps({"sid": sid, "plans": plans}) + "\n"); part.flush()
            if n % 32 == 31:
                print(f"{len(have)}/{count}", flush=True)
    for sid in ids:
        for v, (st, obj, it, evn) in enumerate(have[sid]):
            status[sid - first, v], objective[sid - first, v], iters[sid - first, v], evals[sid - first, v] = st, obj, it, evn
    os.remove(part_path)
    path = os.path.join(ROOT, "tests", "golden", "mc512_cpu_verdicts.npz")
    np.savez_compressed(path, first_id=np.array(first), iteration_cap=np.array(ITERATION_CAP), variants=np.array([(int(m), s) for m, s in VARIANTS]), status=status, objective=objective,
                        iters=iters, evals=evals)
    fails = int(np.sum(status.min(axis=1) < 0))
    print(f"wrote {path}: {count} scenarios, {fails} on which at least one CPU variant fails, {int(np.sum(status.max(axis=1) < 0))} on which all do")


if __name__ == "__main__":
    main()
