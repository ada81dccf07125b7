# round 6: the A/B and probe calls of the round, one per section (bash scripts/r06/gpu_experiments.sh <section>); the record call is gpu_record.sh
case "$1" in
call1)
# round 6, first call: the new tests (reference GPU header through the drop-in, full Monte-Carlo share, device-form failure path), the whole GPU suite, a bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_reference_gpu_header.py tests/test_gpu_parity.py::test_device_form_notices_an_expired_wait_by_itself tests/test_gpu_parity.py::test_one_launch_evaluation_fails_loudly_and_the_handle_stays_usable "tests/test_gpu_configs.py::test_config4_full_share_of_one_gpu" tests/test_takeover.py -m gpu -q -p no:cacheprovider --timeout 800 -s -x > gpurun_out/r06_new_tests.log 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_new_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_new_tests.log | head -30
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 --durations=8 > gpurun_out/r06_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests.log | head -30
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_call1.json 2> gpurun_out/bench1.err; echo "bench rc=$?"; tail -2 gpurun_out/bench1.err | cut -c1-300
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_call1.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate']})
print(json.dumps(d.get('boundary_call_us'))[:1500])
PY
;;
call2)
# round 6, second call: the penalty integrator's instruction diet build against build, its parity tests, the service-protocol probe, a bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_gpu_header.py tests/test_golden.py -m gpu -q -p no:cacheprovider --timeout 500 > gpurun_out/r06_tests2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests2.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests2.log | head -30
timeout 600 python scripts/r06/penalty_ab.py > gpurun_out/r06_penalty_ab.jsonl 2> gpurun_out/penalty_ab.err; echo "penalty ab rc=$?"; cut -c1-420 gpurun_out/r06_penalty_ab.jsonl; tail -3 gpurun_out/penalty_ab.err
timeout 120 scripts/micro/service_probe 200 > gpurun_out/r06_service_probe.jsonl 2>&1; echo "service probe rc=$?"; cat gpurun_out/r06_service_probe.jsonl
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_call2.json 2> gpurun_out/bench2.err; echo "bench rc=$?"; tail -2 gpurun_out/bench2.err | cut -c1-300
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_call2.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate']})
print(json.dumps(d.get('boundary_call_us'))[:1800])
print(json.dumps(d['roofline'].get('large_batch'))[:900])
PY
;;
call3)
# round 6, third call: kernel arguments by pointer (k_eval_cluster, k_round) against by value, alternating processes on one box; parity of the touched paths first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py tests/test_takeover.py -m gpu -q -p no:cacheprovider --timeout 500 > gpurun_out/r06_tests3.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests3.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests3.log | head -30
AB_KAPPA48=1 timeout 1500 python scripts/ab_env.py "FRX_ROUND_ARGPTR=0 FRX_EVAL_ARGPTR=0" "-" 4 > gpurun_out/r06_ab_argptr.jsonl 2> gpurun_out/ab_argptr.err; echo "ab rc=$?"; tail -1 gpurun_out/r06_ab_argptr.jsonl
;;
call4)
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
;;
call5)
# round 6, fifth call: the two modes of a process and where its mailbox pages live - alternating processes with the allocation scope on (default) and off (round 5's behaviour)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 python scripts/r06/mode_probe.py 1 | sed 's/^{/{"numa_alloc": "device node (default)", /' | cut -c1-260
  timeout 100 python scripts/r06/mode_probe.py 1 FRX_NUMA_ALLOC=0 | sed 's/^{/{"numa_alloc": "wherever the caller runs (FRX_NUMA_ALLOC=0)", /' | cut -c1-260
done > gpurun_out/r06_mode_probe_ab.jsonl 2> gpurun_out/mode_ab.err
cat gpurun_out/r06_mode_probe_ab.jsonl | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_multi.py tests/test_takeover.py -m gpu -q -p no:cacheprovider --timeout 500 > gpurun_out/r06_tests5.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests5.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests5.log | head -20
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_call5.json 2> gpurun_out/bench5.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_call5.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','sclk_mhz_under_latency_bound_fp64_load','plan_kilocycles_per_round']})
print(json.dumps(d.get('boundary_call_us'))[:1200])
PY
;;
call6)
# round 6, sixth call: WHERE is host memory on the round's chain?  The round's timeline (instrumented instantiation) with the mailboxes on the device's NUMA node
# and on whatever node the caller ran (several processes: the slow mode shows in most of them), segment by segment
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/r06_gaps_local_$i.txt 2>&1
  FRX_NUMA_ALLOC=0 timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/r06_gaps_anynode_$i.txt 2>&1
done
head -3 gpurun_out/r06_gaps_local_1.txt | cut -c1-200
python - <<'PY'
import re, glob
def load(f):
    rows = {}
    for l in open(f):
        m = re.match(r"\s*(.+?->.+?)\s+mean\s+([0-9.]+)\s+median\s+([0-9.]+)", l)
        if m: rows[re.sub(r"\s+", " ", m.group(1).strip())] = (float(m.group(2)), float(m.group(3)))
    hdr = open(f).readline()
    return rows, hdr
files = sorted(glob.glob("gpurun_out/r06_gaps_*.txt"))
data = {f: load(f) for f in files}
keys = list(data[files[0]][0].keys())
print("segment".ljust(62) + " ".join(f.split("r06_gaps_")[1][:-4].rjust(10) for f in files))
for k in keys:
    print(k[:60].ljust(62) + " ".join(("%.2f" % data[f][0].get(k, (float('nan'),))[0]).rjust(10) for f in files))
for f in files: print(f, data[f][1][:160].strip())
PY
;;
*) echo "sections: call1 .. call6"; exit 2;;
esac
