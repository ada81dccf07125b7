cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
B, N, gates, kappa = sc.CONFIGS["montecarlo4096"]; B //= 8
cands = [sc.make_candidate(b, N, gates) for b in range(B)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
x0 = prob.initial_guess(); tol = sc.ZHANGJIAJIE["opt_rel_tol"]
prob.set_resident(False); a = prob.optimize(tol, x0=x0)
prob.set_resident(2); b = prob.optimize(tol, x0=x0)
bad = np.where((a["status"] < 0) | (b["status"] < 0) | (a["status"] != b["status"]))[0]
print("per-stage ms", a["ms_total"], "queue ms", b["ms_total"])
for i in bad: print("scenario", int(i), "per-stage status", int(a["status"][i]), "evals", int(a["evals"][i]), "obj %.4g" % a["objective"][i], "| queue status", int(b["status"][i]), "evals", int(b["evals"][i]), "obj %.4g" % b["objective"][i])
top = np.argsort(-a["evals"])[:3]; print("most evaluations per-stage:", [(int(i), int(a["evals"][i]), int(a["status"][i])) for i in top], "queue:", [(int(i), int(b["evals"][i]), int(b["status"][i])) for i in np.argsort(-b["evals"])[:3]])
print("median objective", float(np.median(a["objective"])))
PY
