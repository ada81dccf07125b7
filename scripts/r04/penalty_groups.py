#!/usr/bin/env python3
"""(round 4: run once per FRX_PENALTY_GROUPS value by scripts/r04/gpu_pen.sh)  k_penalty_lat with W = 1..4 waves per workgroup (FRX_PENALTY_WAVES) at kappa = 16 and the stock kappa = 48, several batch sizes,
HIP events on the launch stream; checks that every W gives the same out20 as W = 1 (fixed-order reductions: bit-equal)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from frx_import import frx
from fast_racing_amd import scenario as sc

B0, N, gates = 32, 64, 16
stream = torch.cuda.current_stream().cuda_stream
for kappa in (16,):
    base = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B0)]
    os.environ.pop("FRX_PENALTY_WAVES", None)
    p0 = frx.Problem(base, sc.ZHANGJIAJIE, qd_intervals=kappa)
    xs = p0.optimize(1e-6, max_iterations=60)["x"]
    p0.close()
    for B in (32, 1024, 4096):
        rep = B // B0
        ref = None
        for W in ((1,), (4,)) if False else (1, 2, 4):
            os.environ["FRX_PENALTY_WAVES"] = str(W)
            prob = frx.Problem(base * rep, sc.ZHANGJIAJIE, qd_intervals=kappa)
            xb = np.concatenate([xs] * rep)
            T, Cf = prob.forward(xb)
            Td = torch.from_numpy(T).cuda(); Cd = torch.from_numpy(Cf.reshape(-1)).cuda()
            out = torch.zeros(prob.P * 20, dtype=torch.float64, device="cuda")
            for _ in range(5): prob.penalty_device(Td.data_ptr(), Cd.data_ptr(), out.data_ptr(), stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 100 if B <= 256 else 30
            e0.record()
            for _ in range(reps): prob.penalty_device(Td.data_ptr(), Cd.data_ptr(), out.data_ptr(), stream)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            o = out.cpu().numpy()
            if ref is None: ref = o
            print(json.dumps(dict(kappa=kappa, B=B, W=W, pen_us=round(us, 2), hbm_frac=round(prob.algorithmic_bytes() / us / 1e3 / 8000, 4),
                                  Gsamples=round(prob.samples() / us / 1e3, 2), equal_to_W1=bool(np.array_equal(o, ref)))), flush=True)
            prob.close()
            if B == 4096 and W == 1: pass
        if B >= 4096: break
os.environ.pop("FRX_PENALTY_WAVES", None)
