"""How long does a burst of work have to last before the device runs it at its sustained rate?  K = 20 evaluation steps as one hipGraph, replayed back to back
after an idle stretch (the process sleeps 50 ms): microseconds per step of every replay in order.   python scripts/r05/graph_ramp_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from frx_import import frx
from fast_racing_amd import scenario as sc
B, N, gates, kappa = sc.CONFIGS["headline"]
cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
xs = prob.optimize(sc.ZHANGJIAJIE["opt_rel_tol"], max_iterations=60)["x"]
x_dev = torch.from_numpy(xs).cuda(); f_dev = torch.zeros(prob.B, dtype=torch.float64, device="cuda"); g_dev = torch.zeros(prob.NX, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream(); K = 20
out = {}
with torch.cuda.stream(s):
    for _ in range(5): prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), s.cuda_stream)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
    g.replay(); g.replay(); s.synchronize()
    for label, idle in (("after_50ms_idle", 0.05), ("after_50ms_idle_again", 0.05), ("no_idle", 0.0)):
        time.sleep(idle)
        n = 40
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record(s)
        for i in range(n): g.replay(); evs[i + 1].record(s)
        s.synchronize()
        out[label] = [round(evs[i].elapsed_time(evs[i + 1]) * 1e3 / K, 2) for i in range(n)]
print(json.dumps(out))
