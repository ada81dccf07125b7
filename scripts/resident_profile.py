"""Per-segment time of the resident round kernel (FRX_RESIDENT_PROF instantiation): leader, dense and one plain member workgroup of
candidate 0, microseconds per round.  Usage: python scripts/resident_profile.py [B] [N] [kappa] [max_iterations]"""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.environ["FRX_ROOT"]) if os.environ.get("FRX_ROOT") else os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # FRX_ROOT: a variant directory
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kappa = int(sys.argv[3]) if len(sys.argv) > 3 else 16
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 400
SEG = ["wait_host", "vectors", "forward", "wait_phase", "pass_a", "wait_part", "dense_in", "solve", "wait_u", "pass_b", "penalty", "wait_arrive", "gather", "backward", "post", "publish"]
cands = [sc.make_candidate(0, N, N // 4, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
x0 = prob.initial_guess()
prob.optimize(1e-6, x0=x0, max_iterations=20)                     # warm
os.environ["FRX_RESIDENT_PROF"] = os.environ.get("FRX_PROFILE_MODE", "1")      # 2: segment times without the bodies' cycle stamps
r = prob.optimize(1e-6, x0=x0, max_iterations=iters)
del os.environ["FRX_RESIDENT_PROF"]
pr = prob.resident_profile()
rounds = int(r["evals"][0])
G = pr.shape[1]
out = {"B": B, "N": N, "kappa": kappa, "G": G, "rounds_cand0": rounds, "iters_cand0": int(r["iters"][0]), "plan_ms": r["ms_total"], "us_per_round_wall": 1e3 * r["ms_total"] / max(r["rounds"], 1)}
# the three roles re-use the 16 accumulators; the keys keep the history workgroups' names, these say what a key means in the other two loops
out["what the keys mean for the leader"] = {"wait_host": "waiting for a command that was not predicted", "vectors": "trial point (when the gather has not formed it) + barrier",
    "forward": "forward map", "dense_in": "command decoded / accepted step taken over", "solve": "drain + meeting in front of a phase word", "publish": "phase word, deferred result post, accept copies",
    "penalty": "own penalty share (none at the headline geometry)", "wait_arrive": "arrival count (only phases that wait: INIT / NEXT / QUIT, or every phase without granules)",
    "gather": "waiting for the direction granules + gather + first trial point", "backward": "waiting for the penalty partials' granules + adjoint",
    "pass_a": "confirmation of the command this round ran on", "wait_part": "prediction of the next command", "post": "result post"}
out["what the keys mean for the dense workgroup"] = {"wait_phase": "idle", "wait_part": "waiting for the partial sums", "dense_in": "gather of the partial sums", "pass_a": "pass 1",
    "forward": "new column + pass 2", "pass_b": "pass 3", "solve": "granules out", "vectors": "Y^T Y update (off the critical path)"}
for name, wg in (("leader", 0), ("member1", 1), ("dense", G - 1)):
    out[name] = {SEG[i]: round(float(pr[0, wg, i]) / rounds, 2) for i in range(16) if pr[0, wg, i] > 0}
    out[name]["total"] = round(float(pr[0, wg].sum()) / rounds, 2)
out["host_wait_histogram_all_leaders (bin k: < 2^k us)"] = [int(v) for v in prob.last_host_wait_hist.sum(axis=0)]
st = prob.last_stamps
# cycle stamps of candidate 0's last evaluation, relative to the body's entry (wave-specialised bodies of <= 64 pieces):
#   forward: matrix wave (thread 0) = staged, T ready, all steps done, left; axis wave 1 (thread 64) = waypoint map done, met the matrix
#   wave, right-hand side built, last step applied, coefficients stored
#   adjoint: thread 0 = staged, after the axis phase's barrier, time side done, end; thread 64 = start of the axis work, Hermite adjoint done,
#   adjoint solve done, knot adjoint done, waypoint layer: first pass done (30), scalars done (31), done (29)
rel = lambda idx, base: {str(i): int(st[i] - st[base]) for i in idx if st[i] > 0}
out["forward_stamps"] = {"matrix_wave": rel([1, 2, 5, 6], 0), "axis_wave": rel([8, 9, 10, 11, 12], 0), "publication (leader thread 0: before its drain, drained, phase word stored)": rel([13, 14, 15], 0)}
out["adjoint_stamps"] = {"wave0": rel([17, 22, 23, 24], 16), "axis_wave": rel([25, 26, 27, 28, 30, 31, 29], 16)}
if os.environ.get("FRX_RESIDENT_TIMED_READ") == "1" and st[20] > 0:           # experiment: an extra, timed 16-byte read of the command mailbox per round (leader 0)
    out["timed_host_read_us"] = {"mean": float(st[19]) / float(st[20]) / 100.0, "max": float(st[21]) / 100.0, "reads": int(st[20])}
print(json.dumps(out, indent=1))
prob.set_resident(False)
r2 = prob.optimize(1e-6, x0=x0, max_iterations=iters)
print(json.dumps({"per_stage_ms": r2["ms_total"], "per_stage_us_per_round": 1e3 * r2["ms_total"] / r2["rounds"]}))
