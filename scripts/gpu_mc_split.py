"""Monte-Carlo share of one GPU (512 scenarios) as one handle vs several handles driven by concurrent host threads
(each handle owns its stream, so the HBM-bound recursion of one group overlaps the FP64-bound evaluation of another)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frx_import  # noqa
import fast_racing_amd as frx, fast_racing_amd.scenario as sc
B = 512
cands = [sc.make_candidate(b, 64, 16) for b in range(B)]
for groups in (1, 2, 4):
    per = B // groups
    probs = [frx.Problem(cands[i * per:(i + 1) * per], sc.ZHANGJIAJIE, qd_intervals=16) for i in range(groups)]
    x0 = [p.initial_guess() for p in probs]
    res = [None] * groups
    def run(i): res[i] = probs[i].optimize(sc.ZHANGJIAJIE["opt_rel_tol"], x0=x0[i])
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i,)) for i in range(groups)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    ok = sum(int((r["status"] >= 0).sum()) for r in res)
    print(f"groups {groups}: {dt * 1e3:.0f} ms for {B} plans = {B / dt:.0f} plans/s, converged {ok}, rounds {[r['rounds'] for r in res]}")
    [p.close() for p in probs]
