#!/usr/bin/env python3
"""k_penalty_lat2 build against build in ONE process: the round-5 form (FRX_PENALTY_TWOPHASE=5), the one-phase form (=0) and the round-6 form
(default) at 256 / 1024 / 4096 candidates of the headline geometry and 1024 of the stock kappa = 48, alternating, HIP events; outputs compared bit for bit."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from frx_import import frx
from fast_racing_amd import scenario as sc

def run(kappa, batches, reps=40, rounds=5):
    B0, N, gates, _ = sc.CONFIGS["headline"]
    base = [sc.make_candidate(0, N, gates, perturb_id=b) for b in range(B0)]
    p0 = frx.Problem(base, sc.ZHANGJIAJIE, qd_intervals=kappa)
    xs = p0.optimize(1e-6, max_iterations=60)["x"]
    p0.close()
    stream = torch.cuda.current_stream().cuda_stream
    for B in batches:
        rep = B // B0
        prob = frx.Problem(base * rep, sc.ZHANGJIAJIE, qd_intervals=kappa)
        T, Cf = prob.forward(np.concatenate([xs] * rep))
        Td = torch.from_numpy(T).cuda(); Cd = torch.from_numpy(Cf.reshape(-1)).cuda()
        outs, times = {}, {}
        for rnd in range(rounds):
            for form in ("5", "0", "1"):
                os.environ["FRX_PENALTY_TWOPHASE"] = form
                out = torch.zeros(prob.P * 20, dtype=torch.float64, device="cuda")
                for _ in range(5): prob.penalty_device(Td.data_ptr(), Cd.data_ptr(), out.data_ptr(), stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): prob.penalty_device(Td.data_ptr(), Cd.data_ptr(), out.data_ptr(), stream)
                e1.record(); torch.cuda.synchronize()
                times.setdefault(form, []).append(e0.elapsed_time(e1) * 1e3 / reps)
                if rnd == 0: outs[form] = out.cpu().numpy()
        os.environ.pop("FRX_PENALTY_TWOPHASE", None)
        byts = prob.algorithmic_bytes()
        row = {"kappa": kappa, "candidates": B, "algorithmic_bytes": byts,
               "us_round5_two_phase": float(np.median(times["5"])), "us_one_phase": float(np.median(times["0"])), "us_round6_two_phase": float(np.median(times["1"])),
               "frac_round5": byts / np.median(times["5"]) / 1e3 / 8000, "frac_round6": byts / np.median(times["1"]) / 1e3 / 8000,
               "bit_identical_r6_vs_r5": bool(np.array_equal(outs["1"], outs["5"])), "bit_identical_r6_vs_one_phase": bool(np.array_equal(outs["1"], outs["0"])),
               "all_runs_us": {k: [round(v, 2) for v in vs] for k, vs in times.items()}}
        print(json.dumps(row), flush=True)
        prob.close()

run(16, [256, 1024, 4096])
run(48, [1024])
