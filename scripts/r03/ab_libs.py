"""A/B of two BUILDS on one box: alternating subprocesses, each loads the package (and its libfrx.so) from its own root.
   python ab_libs.py ROOT_A ROOT_B [reps] [B]   -> us per round of the full plan, per process"""
import json, os, subprocess, sys
child = r'''
import os, sys, json
root = sys.argv[1]; B = int(sys.argv[2])
sys.path.insert(0, root)
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess()
tol = sc.ZHANGJIAJIE["opt_rel_tol"]
prob.optimize(tol, x0=x0, max_iterations=50)
v = []
for i in range(3):
    r = prob.optimize(tol, x0=x0)
    v.append(round(1e3 * r["ms_total"] / r["rounds"], 3))
print(json.dumps({"us_per_round": v, "rounds": int(r["rounds"]), "objective_min": float(r["objective"].min())}))
'''
ra, rb = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B = sys.argv[4] if len(sys.argv) > 4 else "32"
out = {ra: [], rb: []}
for i in range(reps):
    for root in (ra, rb):
        p = subprocess.run([sys.executable, "-c", child, root, B], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120)
        out[root].append(json.loads(p.stdout.strip().splitlines()[-1]))
print(json.dumps(out))
