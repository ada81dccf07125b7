"""One launch per evaluation (frx_eval_kernel.hpp) against the three stage launches: same numbers?  how long?
For each BASELINE config that the form applies to: f and gradient of both forms at a mid-plan point (bit for bit), the average duration of an evaluation in
each form (HIP events inside the library, 200 back-to-back evaluations), and for the headline batch a hipGraph of K evaluations replayed back to back.
   python scripts/r05/eval_fused_probe.py [config ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from frx_import import frx
from fast_racing_amd import scenario as sc

names = sys.argv[1:] or ["headline", "plumbing", "synthetic8"]
out = {}
for name in names:
    B, N, gates, kappa = sc.CONFIGS[name]
    cands = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    r = {"B": B, "N": N, "kappa": kappa, "workgroups_per_candidate": prob.eval_fused()}
    x0 = prob.initial_guess()
    xs = prob.optimize(sc.ZHANGJIAJIE["opt_rel_tol"], x0=x0, max_iterations=60)["x"]
    if prob.eval_fused():
        worst_f = worst_g = 0.0
        for x in (x0, xs, xs + 1e-3 * np.sin(np.arange(xs.size))):
            prob.set_eval_fused(True); f1, g1 = prob.objective(x)
            f1b, g1b = prob.objective(x)                                   # (a second evaluation: the tags move on)
            prob.set_eval_fused(False); f3, g3 = prob.objective(x)
            worst_f = max(worst_f, float(np.max(np.abs(f1 - f3))), float(np.max(np.abs(f1 - f1b))))
            worst_g = max(worst_g, float(np.max(np.abs(g1 - g3))), float(np.max(np.abs(g1 - g1b))))
        r["max_abs_diff_f"] = worst_f; r["max_abs_diff_g"] = worst_g
        prob.set_eval_fused(True); r["one_launch_us"] = prob.eval_launch_time(xs, 200)
    prob.set_eval_fused(False); r["three_launches_us"] = prob.eval_launch_time(xs, 200)
    r["stage_us"] = prob.stage_times(xs, 200)
    prob.set_eval_fused(True)
    if name == "headline" and prob.eval_fused():
        x_dev = torch.from_numpy(xs).cuda(); f_dev = torch.zeros(prob.B, dtype=torch.float64, device="cuda"); g_dev = torch.zeros(prob.NX, dtype=torch.float64, device="cuda")
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(5): prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), s.cuda_stream)
            s.synchronize()
            g_direct = g_dev.clone()
            for K in (20, 200):
                g = torch.cuda.CUDAGraph(); g_dev.zero_()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(K): prob.objective_device(x_dev.data_ptr(), f_dev.data_ptr(), g_dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
                g.replay(); g.replay(); s.synchronize()
                r[f"graph_K{K}_same_gradient"] = bool(torch.equal(g_dev, g_direct))
                series = []
                for rep in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s); g.replay(); e1.record(s); s.synchronize()
                    series.append(round(e0.elapsed_time(e1) * 1e3 / K, 3))
                r[f"graph_K{K}_us_per_step"] = series
    out[name] = r
print(json.dumps(out))
