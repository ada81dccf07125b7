"""How many one-launch evaluations (frx_eval_kernel.hpp) can run SIDE BY SIDE on one device?  n processes, each with its own handle of the headline batch
(32 candidates x 7 workgroups = 224 of 256 CUs' worth), evaluate in a loop at the same time; a cluster that cannot assemble within the 2 s bound makes its
blocking call fail (FRX_ERR_TIMEOUT) - counted.   python scripts/r05/eval_fused_side_by_side.py [n ...]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B, N, gates, kappa = sc.CONFIGS["headline"]
prob = frx.Problem([sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)], sc.ZHANGJIAJIE, qd_intervals=kappa)
x = prob.initial_guess()
f0, g0 = prob.objective(x)
go = float(sys.argv[1])
while time.time() < go: pass
fails = 0; bad = 0; t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3.0:
    try:
        f, g = prob.objective(x); n += 1
        if not (np.array_equal(f, f0)): bad += 1
    except frx.FrxError:
        fails += 1
        prob.set_eval_fused(1)
print(json.dumps({"evaluations": n, "failed": fails, "wrong": bad, "us_per_blocking_evaluation": 3e6 / max(n, 1), "fused": prob.eval_fused()}))
''' % ROOT
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6]:
    go = time.time() + 25.0
    ps = [subprocess.Popen([sys.executable, "-c", child, str(go)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(n)]
    outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
    print(json.dumps({"processes": n, "evaluations": sum(o["evaluations"] for o in outs), "failed": sum(o["failed"] for o in outs), "wrong": sum(o["wrong"] for o in outs),
                      "us_per_blocking_evaluation_per_process": [round(o["us_per_blocking_evaluation"], 1) for o in outs]}), flush=True)
