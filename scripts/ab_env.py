"""A/B of run-time SWITCHES of one build on one box (boxes differ by 3-5 %, their hosts by more): alternating subprocesses, each with one of the given environments.
Per process: the headline plan (32 candidates) and the one-candidate plan - us per round, rounds, a checksum of the optimised x (bit-identity between variants) - the
evaluation in the form frx_objective_eval_device takes (HIP events around 300 back-to-back launches, best of three) with a checksum of (f, grad), the three stage kernels.
   python scripts/ab_env.py "VAR=a VAR2=b" "VAR=c" ... [reps]       ("-" = the default environment)     -> one JSON line per (variant, repetition), then medians"""
import hashlib, json, os, subprocess, sys
child = r'''
import os, sys, json, hashlib
sys.path.insert(0, sys.argv[1])
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
    out[f"B{B}"] = {"us_per_round": v, "rounds": int(r["rounds"]), "plan_ms": round(r["ms_total"], 2), "x_sha": hashlib.sha1(np.ascontiguousarray(r["x"]).tobytes()).hexdigest()[:12],
                    "status_ok": int((r["status"] >= 0).sum()), "resident": int(r["resident"])}
    if B == 32:
        xs = prob.optimize(tol, x0=x0, max_iterations=60)["x"]
        out["eval_one_launch_us"] = round(min(prob.eval_launch_time(xs, reps=300) for _ in range(3)), 3) if prob.eval_fused() else None
        f, g = prob.objective(xs)
        out["eval_sha"] = hashlib.sha1(np.ascontiguousarray(g).tobytes() + np.ascontiguousarray(f).tobytes()).hexdigest()[:12]
        st = [prob.stage_times(xs, reps=300) for _ in range(2)]
        out["stage_us"] = {k: round(min(s[k] for s in st), 3) for k in st[0]}
    prob.close()
if os.environ.get("AB_KAPPA48"):
    cands = [sc.make_candidate(0, 64, 16, perturb_id=0)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=48)
    x0 = prob.initial_guess()
    prob.optimize(tol, x0=x0, max_iterations=50)
    r = prob.optimize(tol, x0=x0)
    out["plumbing_k48"] = {"us_per_round": round(1e3 * r["ms_total"] / r["rounds"], 3), "rounds": int(r["rounds"]), "plan_ms": round(r["ms_total"], 2), "resident": int(r["resident"])}
    prob.close()
out["sclk_mhz"] = [round(v, 1) for v in frx.shader_clock(0, 3.0)]
print(json.dumps(out))
'''
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
variants = [a for a in args if not a.isdigit()]
reps = int([a for a in args if a.isdigit()][0]) if any(a.isdigit() for a in args) else 3
res = {v: [] for v in variants}
for i in range(reps):
    for v in variants:
        env = dict(os.environ)
        if v != "-":
            for kv in v.split(): k, _, val = kv.partition("="); env[k] = val
        try:
            p = subprocess.run([sys.executable, "-c", child, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
            d = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as e:
            d = {"error": repr(e), "stderr": (p.stderr[-400:] if "p" in dir() else "")}
        d["variant"] = v; d["rep"] = i
        res[v].append(d); print(json.dumps(d), flush=True)
import statistics as st
summ = {}
for v, rows in res.items():
    ok = [r for r in rows if "error" not in r]
    if not ok: continue
    summ[v] = {"B32_us_per_round_median": st.median(x for r in ok for x in r["B32"]["us_per_round"]), "B32_rounds": sorted(set(r["B32"]["rounds"] for r in ok)), "B32_x_sha": sorted(set(r["B32"]["x_sha"] for r in ok)),
               "B1_us_per_round_median": st.median(x for r in ok for x in r["B1"]["us_per_round"]), "B1_rounds": sorted(set(r["B1"]["rounds"] for r in ok)),
               "eval_one_launch_us_median": st.median(r["eval_one_launch_us"] for r in ok) if ok[0].get("eval_one_launch_us") else None, "eval_sha": sorted(set(r["eval_sha"] for r in ok)),
               "stage_us_median": {k: st.median(r["stage_us"][k] for r in ok) for k in ok[0]["stage_us"]}}
    if "plumbing_k48" in ok[0]: summ[v]["plumbing_k48_us_per_round_median"] = st.median(r["plumbing_k48"]["us_per_round"] for r in ok)
print(json.dumps({"summary": summ}))
