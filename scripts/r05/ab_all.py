"""Round-5 A/B of BUILDS on one box (boxes differ by 3-5 %): alternating subprocesses, each loads the package and its libfrx.so from its own root
(a variant directory made by scripts/r04/make_variant.sh, or the repository itself).  Per process: the headline plan (32 candidates) and the
one-candidate plan - us per round, rounds, a checksum of the optimised x (bit-identity between builds) - the three stage kernels one at a time
(frx_eval_stage_times), the penalty integrator on 1024 candidates, and optionally the kappa = 48 plan of BASELINE configs[0].
   python scripts/r05/ab_all.py ROOT_A ROOT_B ... [reps]        -> one JSON line per (root, repetition), then a summary"""
import hashlib, json, os, subprocess, sys
child = r'''
import os, sys, json, hashlib
root = sys.argv[1]
sys.path.insert(0, root)
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
out = {}
tol = sc.ZHANGJIAJIE["opt_rel_tol"]
for B in (32, 1):
    cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
    x0 = prob.initial_guess()
    prob.optimize(tol, x0=x0, max_iterations=50)
    v = []
    for i in range(3 if B == 32 else 2):
        r = prob.optimize(tol, x0=x0)
        v.append(round(1e3 * r["ms_total"] / r["rounds"], 3))
    out[f"B{B}"] = {"us_per_round": v, "rounds": int(r["rounds"]), "plan_ms": round(r["ms_total"], 2), "objective_min": float(r["objective"].min()),
                    "x_sha": hashlib.sha1(np.ascontiguousarray(r["x"]).tobytes()).hexdigest()[:12], "status_ok": int((r["status"] >= 0).sum())}
    if B == 32:
        xs = prob.optimize(tol, x0=x0, max_iterations=60)["x"]
        st = [prob.stage_times(xs, reps=300) for _ in range(2)]
        out["stage_us"] = {k: round(min(s[k] for s in st), 3) for k in st[0]}
        f, g = prob.objective(xs)
        out["eval_sha"] = hashlib.sha1(np.ascontiguousarray(g).tobytes() + np.ascontiguousarray(f).tobytes()).hexdigest()[:12]
        big = frx.Problem(cands * 32, sc.ZHANGJIAJIE, qd_intervals=16)
        sb = [big.stage_times(np.tile(xs, 32), reps=40) for _ in range(2)]
        out["penalty_us_1024"] = round(min(s["penalty"] for s in sb), 2)
        big.close()
    prob.close()
if os.environ.get("AB_KAPPA48"):
    cands = [sc.make_candidate(0, 64, 16, perturb_id=0)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=48)
    x0 = prob.initial_guess()
    prob.optimize(tol, x0=x0, max_iterations=50)
    r = prob.optimize(tol, x0=x0)
    out["plumbing_k48"] = {"us_per_round": round(1e3 * r["ms_total"] / r["rounds"], 3), "rounds": int(r["rounds"]), "plan_ms": round(r["ms_total"], 2), "resident": int(r["resident"])}
    prob.close()
print(json.dumps(out))
'''
args = sys.argv[1:]
roots = [os.path.abspath(a) for a in args if not a.isdigit()]
reps = int([a for a in args if a.isdigit()][0]) if any(a.isdigit() for a in args) else 3
res = {r: [] for r in roots}
for i in range(reps):
    for root in roots:
        try:
            env = dict(os.environ); env["FRX_RESIDENT_HOST_STATS"] = "1"      # per plan on stderr: clusters on one XCD, scans per mailbox thread
            p = subprocess.run([sys.executable, "-c", child, root], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240, env=env)
            d = json.loads(p.stdout.strip().splitlines()[-1])
            import re
            d["one_xcd"] = [int(a) for a, b in re.findall(r"clusters on one XCD: (\d+) of (\d+)", p.stderr)]
            d["us_per_scan_thread0"] = [float(x) for x in re.findall(r"mailbox thread 0: .*? = ([0-9.]+) us per scan", p.stderr)]
        except Exception as e:
            d = {"error": repr(e), "stderr": (p.stderr[-400:] if 'p' in dir() else "")}
        res[root].append(d)
        print(json.dumps({"root": os.path.basename(root), "rep": i, **d}), flush=True)
import numpy as np
print("---- summary (median over processes of the per-process medians) ----")
for root, v in res.items():
    ok = [d for d in v if "B32" in d]
    if not ok: print(os.path.basename(root), "no result"); continue
    med = lambda f: round(float(np.median([f(d) for d in ok])), 3)
    print(json.dumps({"root": os.path.basename(root), "B32_us_per_round": med(lambda d: np.median(d["B32"]["us_per_round"])), "B32_rounds": ok[0]["B32"]["rounds"], "B32_plan_ms": med(lambda d: d["B32"]["plan_ms"]),
                      "B1_us_per_round": med(lambda d: np.median(d["B1"]["us_per_round"])), "B1_rounds": ok[0]["B1"]["rounds"],
                      "stage_us": {k: med(lambda d, k=k: d["stage_us"][k]) for k in ok[0]["stage_us"]}, "penalty_us_1024": med(lambda d: d["penalty_us_1024"]),
                      "x_sha_B32": ok[0]["B32"]["x_sha"], "x_sha_B1": ok[0]["B1"]["x_sha"], "eval_sha": ok[0]["eval_sha"],
                      **({"k48": ok[0].get("plumbing_k48")} if "plumbing_k48" in ok[0] else {})}))
