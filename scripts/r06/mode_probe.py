"""The headline plan's two MODES (25.1 / 26.5 us per round, constant within a process, profiles/r06_ab_host_scan.jsonl): what differs between the processes?
N processes, each: three plans, the XCD census of its clusters (FRX_RESIDENT_HOST_STATS), its mailbox scan period, the sustained shader clock."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
from frx_import import frx
from fast_racing_amd import scenario as sc
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(32)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess()
prob.optimize(1e-6, x0=x0, max_iterations=50)
v = []
for i in range(3):
    r = prob.optimize(1e-6, x0=x0)
    v.append(round(1e3 * r["ms_total"] / r["rounds"], 3))
print(json.dumps({"us_per_round": v, "sclk": [round(x, 1) for x in frx.shader_clock(0, 2.0)]}))
'''
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for i in range(n):
    env = dict(os.environ); env["FRX_RESIDENT_HOST_STATS"] = "1"
    for kv in sys.argv[2:]:
        k, _, val = kv.partition("="); env[k] = val
    p = subprocess.run([sys.executable, "-c", child, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    d["one_xcd"] = [int(a) for a, b in re.findall(r"clusters on one XCD: (\d+) of (\d+)", p.stderr)]
    d["us_per_scan"] = [float(x) for x in re.findall(r"mailbox thread \d: .*? = ([0-9.]+) us per scan", p.stderr)][-2:]
    place = re.findall(r"clusters on one XCD: \d+ of \d+;(.*)", p.stderr)
    if place:
        # XCD of the leader of every cluster, and how many distinct (XCD, SE, SA, CU) the leaders use
        cl = re.findall(r"\[([^\]]*)\]", place[-1])
        d["leader_xcd"] = "".join(c.split()[0].split(":")[0] for c in cl)
        d["cu_slots_used_twice"] = len([1 for c in cl for w in c.split()]) - len(set(w for c in cl for w in c.split()))
    print(json.dumps(d), flush=True)
