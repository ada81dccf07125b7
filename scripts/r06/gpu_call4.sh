# round 6, fourth call: is the HOST on the round's chain?  The mailbox threads' scan period stretched on purpose (FRX_RESIDENT_SCAN_PAUSE = extra pause instructions
# between two scans of a thread's mailboxes) against the default, alternating processes; with the box's sustained shader clock next to every line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python scripts/ab_env.py "-" "FRX_RESIDENT_SCAN_PAUSE=20" "FRX_RESIDENT_SCAN_PAUSE=100" "FRX_RESIDENT_SCAN_PAUSE=400" "FRX_RESIDENT_HOST_THREADS=1" 3 > gpurun_out/r06_ab_host_scan.jsonl 2> gpurun_out/ab_host.err; echo "ab rc=$?"; tail -1 gpurun_out/r06_ab_host_scan.jsonl
FRX_RESIDENT_HOST_STATS=1 timeout 100 python - <<'PY' 2>&1 | grep -E "mailbox thread|us_per_round" | head -12
import os, sys, json
sys.path.insert(0, os.getcwd())
from frx_import import frx
from fast_racing_amd import scenario as sc
cands = [sc.make_candidate(0, 64, 16, perturb_id=b) for b in range(32)]
prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
x0 = prob.initial_guess()
for pause in ("0", "20", "100", "400"):
    os.environ["FRX_RESIDENT_SCAN_PAUSE"] = pause
    r = prob.optimize(1e-6, x0=x0)
    print(json.dumps({"scan_pause": pause, "us_per_round": 1e3 * r["ms_total"] / r["rounds"], "rounds": int(r["rounds"])}), flush=True)
PY
