# round 5, call 2: the tests that failed in call 1 (fixed) + the host-budget test; the adjoint's in-round gain on ONE box (timeline statistics of the build
# without and with the restructured adjoint); build against build with more processes; the throughput and the two small BASELINE configs as they are now
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_resident.py "tests/test_gpu_parity.py::test_lockstep_parity_along_the_whole_optimisation" -m gpu -q -p no:cacheprovider --timeout 600 -s > gpurun_out/tests2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests2.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests2.log | head -30
grep -E "^\{\"(us_per_round_lone|resident_vs_per_stage|kappa|rounds_on_predicted)" gpurun_out/tests2.log | cut -c1-700
for v in ab_pen . ab_pen .; do FRX_ROOT=$v timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/r05b_gaps_B32_$(basename $v | sed 's/^\.$/repo/')_$RANDOM.txt 2>&1; done
for f in gpurun_out/r05b_gaps_B32_*; do echo $f; head -1 $f | cut -c1-200; grep -E "forward starts -> forward done|member sees CT|penalty done|drained ->" $f; done
AB_KAPPA48=1 timeout 1200 python scripts/r05/ab_all.py ab_r04 ab_pen . 5 > gpurun_out/ab2.jsonl 2> gpurun_out/ab2.err; tail -4 gpurun_out/ab2.jsonl | cut -c1-900
timeout 400 python bench.py --config montecarlo4096 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline > gpurun_out/r05b_bench_montecarlo4096.json 2> gpurun_out/bench_mc.err; echo "mc rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r05b_bench_montecarlo4096.json').read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('plan') and not isinstance(v,(list,str))})"
for c in plumbing synthetic8; do timeout 300 python bench.py --config $c --steps 200 --warmup 20 > gpurun_out/r05b_bench_$c.json 2> gpurun_out/bench_$c.err; echo "$c rc=$?"; tail -1 gpurun_out/bench_$c.err | cut -c1-300; done
python - <<'PY'
import json
for c in ('plumbing','synthetic8'):
    try:
        d=json.loads(open(f'gpurun_out/r05b_bench_{c}.json').read().strip().splitlines()[-1]); r=d['roofline']
        print(c, {k: d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_per_stage_path','plan_ms_with_setup','plan_path']}, r['stage_kernels_us'], 'frac', r['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('plan_ms_one_candidate_1thread'))
    except Exception as e: print(c, 'failed', e)
PY
