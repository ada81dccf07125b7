"""A/B of two BUILDS on one box: alternating subprocesses, each loads the package (and its libfrx.so) from its own root.
   python ab_libs.py ROOT_A ROOT_B [ROOT_C ...] [reps] [B]   -> us per round of the full plan, per process"""
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
args = sys.argv[1:]
roots = [os.path.abspath(a) for a in args if not a.isdigit()]
nums = [a for a in args if a.isdigit()]
reps = int(nums[0]) if nums else 2
B = nums[1] if len(nums) > 1 else "32"
out = {r: [] for r in roots}
for i in range(reps):
    for root in roots:
        p = subprocess.run([sys.executable, "-c", child, root, B], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120)
        out[root].append(json.loads(p.stdout.strip().splitlines()[-1]))
print(json.dumps(out))
