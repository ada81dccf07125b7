#!/usr/bin/env python3
"""k_penalty alone at several batch sizes and optimisation states (HIP events on the launch stream).
Large batches replicate the 32 headline candidates (every replica owns its own copy of coefficients and
polytopes in HBM, so the traffic is real)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from frx_import import frx
from fast_racing_amd import scenario as sc

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="32,256,1024,4096")
ap.add_argument("--states", default="init,it60,conv")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--full", action="store_true", help="also time the full 3-kernel evaluation")
args = ap.parse_args()

B0, N, gates, kappa = sc.CONFIGS["headline"]
base = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B0)]
p0 = frx.Problem(base, sc.ZHANGJIAJIE, qd_intervals=kappa)
x0 = p0.initial_guess()
states = {"init": x0}
if "it60" in args.states: states["it60"] = p0.optimize(1e-6, x0=x0, max_iterations=60)["x"]
if "conv" in args.states: states["conv"] = p0.optimize(1e-6, x0=x0)["x"]
stream = torch.cuda.current_stream().cuda_stream
rows = []
for B in [int(b) for b in args.batches.split(",")]:
    rep = B // B0
    prob = frx.Problem(base * rep, sc.ZHANGJIAJIE, qd_intervals=kappa) if rep > 1 else p0
    for name in args.states.split(","):
        xs = states[name]
        xb = np.concatenate([xs] * rep) if rep > 1 else xs
        T, Cf = prob.forward(xb)
        Td = torch.from_numpy(T).cuda(); Cd = torch.from_numpy(Cf.reshape(-1)).cuda()
        out = torch.zeros(prob.P * 20, dtype=torch.float64, device="cuda")
        for _ in range(5): prob.penalty_device(Td.data_ptr(), Cd.data_ptr(), out.data_ptr(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps): prob.penalty_device(Td.data_ptr(), Cd.data_ptr(), out.data_ptr(), stream)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        row = dict(B=B, state=name, pen_us=round(us, 2), GBs=round(prob.algorithmic_bytes() / us / 1e3, 1),
                   frac=round(prob.algorithmic_bytes() / us / 1e3 / 8000, 4), Gsamples=round(prob.samples() / us / 1e3, 2))
        if args.full:
            xd = torch.from_numpy(xb).cuda(); fd = torch.zeros(prob.B, dtype=torch.float64, device="cuda"); gd = torch.zeros(prob.NX, dtype=torch.float64, device="cuda")
            # both forms of an evaluation where the one-launch form applies (k_eval_cluster, batches the chip holds at once): the counter passes need the stage
            # kernels AND the cluster kernel at the headline batch
            prob.set_eval_solo(0)                                   # (the stage kernels at every size: the counter passes need them; the solo launch separately below)
            for form in ((False, True) if prob.eval_fused() else (False,)):
                prob.set_eval_fused(form)
                for _ in range(5): prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), stream)
                e0.record()
                for _ in range(args.reps): prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), stream)
                e1.record(); torch.cuda.synchronize()
                row["eval_one_launch_us" if form else "eval_us"] = round(e0.elapsed_time(e1) * 1e3 / args.reps, 2)
            prob.set_eval_fused(True)
            if prob.solo_applies():                                 # one workgroup per candidate, one launch (frx_solo_kernel.hpp)
                prob.set_eval_solo(2)
                for _ in range(5): prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), stream)
                e0.record()
                for _ in range(args.reps): prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), stream)
                e1.record(); torch.cuda.synchronize()
                row["eval_solo_us"] = round(e0.elapsed_time(e1) * 1e3 / args.reps, 2)
            prob.set_eval_solo(1)
        rows.append(row); print(json.dumps(row), flush=True)
    if rep > 1: prob.close()
