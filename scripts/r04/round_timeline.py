"""Timeline of a few phases of the resident round kernel, cluster 0: who does what when (FRX_RESIDENT_PROF=2 + FRX_RESIDENT_TRACE, csrc/frx_api.cpp).
   python scripts/r04/round_timeline.py [B] [phase_lo] [n_phases]  ->  one line per event: t (us), workgroup, role, segment, phase"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frx_import import frx
from fast_racing_amd import scenario as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
nph = int(sys.argv[3]) if len(sys.argv) > 3 else 8
SEG = ["wait_host", "vectors", "forward", "wait_phase", "pass_a", "wait_part", "dense_in", "solve", "wait_u", "pass_b", "penalty", "wait_arrive", "gather", "backward", "post", "publish"]
EXTRA = {32: "chunk_loaded", 33: "dots_done", 34: "products_in_lds", 35: "column_sums_done", 36: "cntL_added", 37: "leader_cntL_added", 40: "loop top", 41: "before the barrier in front of the forward map"}
# what a segment id means at the END of which the event is logged, per role (the macro is shared)
LEADER = {0: "command ready", 1: "trial point formed", 2: "forward map done", 6: "command decoded", 7: "stores drained + met", 15: "phase word out (+ deferred post, accept copies)", 10: "own penalty share done",
          11: "arrival wait over (a no-op in direction / evaluation phases whose data come as granules)", 12: "direction granules polled + gathered + trial point", 13: "penalty partials polled + adjoint done", 4: "command confirmed", 5: "next command predicted", 14: "result posted / round closed"}
MEMBER = {3: "phase word seen", 4: "pass A: partials out, cntA added", 8: "own coefficients (-u, gamma w) polled", 9: "pass B: direction granules stored", 10: "penalty share done"}
DENSE = {3: "phase word seen", 5: "all partials in (cntA)", 6: "partials gathered", 4: "pass 1 done", 2: "column + pass 2 done", 9: "pass 3 done", 7: "-u, gamma w granules stored", 1: "YtY updated"}
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess()
prob.optimize(1e-6, x0=x0, max_iterations=20)
path = os.path.abspath("gpurun_out/round_trace_raw.txt") if os.path.isdir("gpurun_out") else "/tmp/round_trace_raw.txt"
os.environ["FRX_RESIDENT_PROF"] = "2"; os.environ["FRX_RESIDENT_TRACE"] = f"{lo},{lo + nph},{path}"
r = prob.optimize(1e-6, x0=x0, max_iterations=3000)
G = r["resident"]
ev = []
for line in open(path):
    if line.startswith("#"): continue
    w, seg, ph, tk = (int(v) for v in line.split())
    ev.append((tk, w, seg, ph))
ev.sort()
t0 = ev[0][0]
print(f"# B={B} G={G} phases [{lo},{lo + nph}); us per round of this (instrumented) plan: {1e3 * r['ms_total'] / r['rounds']:.2f}")
for tk, w, seg, ph in ev:
    role = "leader" if w == 0 else "dense" if w == G - 1 else f"member{w}"
    names = LEADER if w == 0 else DENSE if w == G - 1 else MEMBER
    name = EXTRA.get(seg) or names.get(seg) or SEG[seg]
    if 1 < w < G - 1 and w != 3: continue                       # members 1 and 3 stand for the six
    print(f"{(tk - t0) / 100.0:9.2f} us  ph {ph:5d}  {role:8s} {name}")
