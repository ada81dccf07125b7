"""Leader's wait for a host command (PROF instantiation) against batch size, poll pause of the idle workgroups and mailbox service threads."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
N, kappa = 64, 16
def run(B, poll=None, threads=None, spec=None, pause=None):
    for k, v in (("FRX_RESIDENT_POLL", poll), ("FRX_RESIDENT_HOST_THREADS", threads), ("FRX_RESIDENT_SPECULATE", spec), ("FRX_RESIDENT_SCAN_PAUSE", pause)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    cands = [sc.make_candidate(0, N, N // 4, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    x0 = prob.initial_guess()
    prob.optimize(1e-6, x0=x0, max_iterations=20)
    os.environ["FRX_RESIDENT_PROF"] = "1"
    r = prob.optimize(1e-6, x0=x0, max_iterations=1500)
    del os.environ["FRX_RESIDENT_PROF"]
    pr = prob.resident_profile()
    rounds = int(r["evals"][0])
    h = prob.last_host_wait_hist.sum(axis=0)
    print(json.dumps({"B": B, "poll": poll, "threads": threads, "speculate": spec, "scan_pause": pause, "us_per_round": round(1e3 * r["ms_total"] / r["rounds"], 2), "wait_host": round(float(pr[0, 0, 0]) / rounds, 2),
                      "post": round(float(pr[0, 0, 14]) / rounds, 2), "hist(<2^k us)": [int(v) for v in h[:9]]}), flush=True)
    prob.close()
if len(sys.argv) > 1 and sys.argv[1] == "pause":
    for pz in (0, 4, 16, 64, 256): run(32, pause=pz)
    for pz in (0, 16, 64): run(32, pause=pz, threads=1)
    sys.exit(0)
for B in (1, 8, 16, 24, 32): run(B)
for poll in (0, 2, 3, 4): run(32, poll=poll)
for th in (1, 4, 8): run(32, threads=th)
run(32, spec=0); run(1, spec=0)
