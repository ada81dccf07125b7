"""Where does test_rccl_exchange_with_every_visible_device spend its time? (call 2: 283 s on one box, 6 s on another)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
t0 = time.perf_counter()
cands = [sc.make_candidate(0, 32, 8, perturb_id=b) for b in range(3)] + [sc.make_candidate(170, 64, 16)]
t1 = time.perf_counter()
mp = frx.MultiProblem(cands, sc.ZHANGJIAJIE, qd_intervals=8)
t2 = time.perf_counter()
r = mp.optimize(1e-5)
t3 = time.perf_counter()
print("scenario %.2fs create %.2fs (rccl %s) optimize %.2fs" % (t1 - t0, t2 - t1, mp.uses_rccl, t3 - t2), "status", r["status"], "iters", r["iters"], "evals", r["evals"], "exchange", r["exchange"])
mp.close()
p = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=8)
t4 = time.perf_counter()
q = p.optimize(1e-5)
t5 = time.perf_counter()
print("single handle optimize %.2fs" % (t5 - t4), "resident", q["resident"], "status", q["status"], "iters", q["iters"], "rounds", q["rounds"], "ms", q["ms_total"], "dev", q["device_status"])
p.set_resident(False)
t4 = time.perf_counter()
q = p.optimize(1e-5)
t5 = time.perf_counter()
print("per-stage optimize %.2fs" % (t5 - t4), "status", q["status"], "iters", q["iters"], "rounds", q["rounds"], "ms", q["ms_total"])
