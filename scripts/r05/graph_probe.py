"""Does a hipGraph of K evaluation steps (3 K kernel nodes, one launch) close the gaps between the stage kernels?  Direct launches against graph replay,
HIP events on the launch stream, headline batch.   python scripts/r05/graph_probe.py [K ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from frx_import import frx
from fast_racing_amd import scenario as sc
Ks = [int(a) for a in sys.argv[1:]] or [20, 200]
B, N, gates, kappa = sc.CONFIGS["headline"]
cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
x0 = prob.initial_guess()
xs = prob.optimize(sc.ZHANGJIAJIE["opt_rel_tol"], x0=x0, max_iterations=60)["x"]
x_dev = torch.from_numpy(xs).cuda(); f_dev = torch.zeros(prob.B, dtype=torch.float64, device="cuda"); g_dev = torch.zeros(prob.NX, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
out = {"stage_us": prob.stage_times(xs, reps=200)}
def ev(): return torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    for _ in range(20): prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), s.cuda_stream)
    s.synchronize()
    g_direct = g_dev.clone()
    for K in Ks:
        best = 1e9
        for rep in range(5):
            e0, e1 = ev(), ev()
            e0.record(s)
            for _ in range(K): prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), s.cuda_stream)
            e1.record(s); s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / K)
        out[f"direct_K{K}_us_per_step"] = best
    for K in Ks:
        try:
            g = torch.cuda.CUDAGraph()
            g_dev.zero_()
            with torch.cuda.graph(g, stream=s):
                for _ in range(K): prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
            g.replay(); s.synchronize()
            same = bool(torch.equal(g_dev, g_direct))
            best = 1e9; series = []
            for rep in range(8):
                e0, e1 = ev(), ev()
                e0.record(s); g.replay(); e1.record(s); s.synchronize()
                series.append(round(e0.elapsed_time(e1) * 1e3 / K, 3))
                best = min(best, series[-1])
            out[f"graph_K{K}_us_per_step"] = best; out[f"graph_K{K}_same_gradient"] = same; out[f"graph_K{K}_replays_in_order"] = series
        except Exception as e:
            out[f"graph_K{K}_error"] = repr(e)[:300]
print(json.dumps(out))
