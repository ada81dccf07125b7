#!/usr/bin/env python3
"""Where do a resident plan's mailboxes live, and does it decide the round's time?  N processes; each loads torch FIRST when asked (as bench.py does: the HIP runtime's host-memory
pools then exist before the library's first pinned allocation), plans the headline batch three times and reports us per round next to the NUMA node of its mailbox pages.
  python scripts/r06/numa_probe.py [processes] [torch|notorch] [VAR=value ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
if sys.argv[2] == "torch":
    import torch
    torch.zeros(1 << 20, device="cuda"); torch.zeros(1 << 16).pin_memory(); torch.cuda.synchronize()
from frx_import import frx
from fast_racing_amd import scenario as sc
import numpy as np
if sys.argv[2] == "torch": frx.dv_selftest(641, B=32, m=128, iters=8)      # (bench.py drives k_lbfgs_pre alone before its plans: pinned command / result buffers come and go)
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(32)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess()
prob.optimize(1e-6, x0=x0, max_iterations=50)
v = []
for i in range(3):
    r = prob.optimize(1e-6, x0=x0)
    v.append(round(1e3 * r["ms_total"] / r["rounds"], 3))
print(json.dumps({"us_per_round": v, **prob.mailbox_numa()}))
'''
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "torch"
for i in range(n):
    env = dict(os.environ)
    for kv in sys.argv[3:]:
        k, _, val = kv.partition("="); env[k] = val
    p = subprocess.run([sys.executable, "-c", child, ROOT, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    try: d = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception: d = {"error": p.stderr[-500:]}
    d["loaded_first"] = mode
    print(json.dumps(d), flush=True)
