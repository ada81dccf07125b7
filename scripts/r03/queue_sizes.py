"""Work queue of the resident round kernel against the per-stage path, Monte-Carlo scenarios (BASELINE configs[4]: independent scenarios,
64 pieces, kappa 16), batch sizes around and beyond the chip's 32 clusters.  One JSON line per size -> gpurun_out/queue_sizes.jsonl."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frx_import import frx                                   # noqa: E402
from fast_racing_amd import scenario as sc                   # noqa: E402

sizes = [int(v) for v in (sys.argv[1:] or ["32", "33", "40", "64", "96", "128", "256", "512"])]
B0, N, gates, kappa = sc.CONFIGS["montecarlo4096"]
tol = sc.ZHANGJIAJIE["opt_rel_tol"]
cands_all = [sc.make_candidate(b, N, gates) for b in range(max(sizes))]
out = open(os.path.join(ROOT, "gpurun_out", "queue_sizes.jsonl"), "a")
for B in sizes:
    prob = frx.Problem(cands_all[:B], sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = prob.initial_guess()
    row = {"candidates": B}
    res = {}
    for name, mode in (("work_queue", 2), ("per_stage", 0)):
        prob.set_resident(mode)
        t0 = time.perf_counter()
        r = prob.optimize(tol, x0=x0, max_iterations=60000)
        wall = (time.perf_counter() - t0) * 1e3
        res[name] = r
        row[name] = {"plan_ms": r["ms_total"], "wall_ms": wall, "plans_per_s": 1e3 * B / r["ms_total"], "resident": int(r["resident"]), "clusters": int(r["clusters"]),
                     "rounds": int(r["rounds"]), "failed": int(np.sum(r["status"] < 0)), "device_status": int(r["device_status"])}
    prob.set_resident(1)
    r = prob.optimize(tol, x0=x0, max_iterations=60000)
    row["default_path"] = "work queue" if r["resident"] and r["clusters"] < B else "resident" if r["resident"] else "per stage"
    row["evals_mean"] = float(res["work_queue"]["evals"].mean()); row["evals_max"] = int(res["work_queue"]["evals"].max())
    row["status_equal"] = bool(np.array_equal(res["work_queue"]["status"], res["per_stage"]["status"]))
    print(json.dumps(row), flush=True)
    out.write(json.dumps(row) + "\n"); out.flush()
    prob.close()
