"""Statistics of the resident round's critical path over ~100 rounds of cluster 0 (timeline trace, see round_timeline.py): mean / median / p90 of the
time between consecutive milestones.   python scripts/r04/round_gaps.py [B] [phase_lo] [n_phases]"""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.environ["FRX_ROOT"]) if os.environ.get("FRX_ROOT") else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # FRX_ROOT: a variant directory (scripts/r04/make_variant.sh)
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
nph = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess()
prob.optimize(1e-6, x0=x0, max_iterations=20)
path = os.path.abspath("gpurun_out/round_trace_raw.txt") if os.path.isdir("gpurun_out") else "/tmp/round_trace_raw.txt"
os.environ["FRX_RESIDENT_PROF"] = "2"; os.environ["FRX_RESIDENT_TRACE"] = f"{lo},{lo + nph},{path}"
r = prob.optimize(1e-6, x0=x0, max_iterations=3000)
G = r["resident"]
ev = {}
for line in open(path):
    if line.startswith("#"): continue
    w, seg, ph, tk = (int(v) for v in line.split())
    ev.setdefault(w, []).append((tk, seg, ph))
L = sorted(ev[0]); M = sorted(ev[1]); D = sorted(ev[G - 1])
def times(evs, seg): return {ph: tk for tk, s, ph in evs if s == seg}      # last event of that kind per phase number
# milestones of an ACCEPTED round: ADV phase p (even/odd unknown) followed by CT phase p + 1
lead = {s: {} for s in range(64)}
for tk, s, ph in L: lead[s].setdefault(ph, tk)
mem = {s: {} for s in range(64)}
for tk, s, ph in M: mem[s].setdefault(ph, tk)
den = {s: {} for s in range(64)}
for tk, s, ph in D: den[s].setdefault(ph, tk)
rows = []
for p in sorted(mem[32]):                                      # phases in which member 1 loaded a chunk = ADVANCE phases
    q = p + 1                                                   # the evaluation phase of the same command
    try:
        m = {"adj_end(prev)": lead[13][p - 1], "confirmed": lead[4][p - 1], "predicted": lead[5][p - 1], "member sees ADV": mem[3][p], "chunk loaded": mem[32][p], "dots done": mem[33][p], "partials out": mem[4][p],
             "dense has all": den[5][p], "dense gathered": den[6][p], "pass1": den[4][p], "pass2": den[2][p], "pass3": den[9][p], "u,w granules out": den[7][p], "member has its coefficients": mem[8][p],
             "products": mem[34][p], "colsums": mem[35][p], "direction granules out": mem[9][p], "leader has direction + trial point": lead[12][p], "forward starts": lead[1][p],
             "forward done": lead[2][p], "drained": lead[7][p], "member sees CT": mem[3][q], "penalty done": mem[10][q], "adj_end (partials polled + adjoint)": lead[13][q]}
    except KeyError:
        continue
    rows.append(m)
names = list(rows[0].keys())
A = np.array([[row[n] for n in names] for row in rows], dtype=np.float64) / 100.0
d = np.diff(A, axis=1)
print(json.dumps({"B": B, "rounds_in_the_sample": len(rows), "us_per_round_instrumented": 1e3 * r["ms_total"] / r["rounds"], "round_us_mean (adjoint end to adjoint end)": float((A[:, -1] - A[:, 0]).mean())}))
for i, n in enumerate(names[1:]):
    print(f"{names[i]:>34s} -> {n:<34s} mean {d[:, i].mean():6.2f}  median {np.median(d[:, i]):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f}  max {d[:, i].max():6.2f}")
