# round 6 record call: parity suite, counter passes (HBM traffic, VALU, dynamic FP64 counts), bench lines (headline: 200 steps and the driver's 20-step form; the other
# BASELINE configs), rocprofv3 kernel statistics of the headline command, round budgets and timeline statistics - ONE call on ONE tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r06}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 400 --durations=5 -s > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests.log | head -20
grep -E "^\{\"(us_per_round_lone|resident_vs_per_stage|kappa|B\"|candidates\")" gpurun_out/tests.log | cut -c1-500 > gpurun_out/${TAG}_test_measurements.jsonl; wc -l gpurun_out/${TAG}_test_measurements.jsonl
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/r06/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log | cut -c1-300
cd $GRAFT_REPO_ROOT
cp gpurun_out/r06_pmc_headline.json profiles/r06_pmc_headline.json            # the bench lines below read the counters of THIS call ("from_profile")
for b in 1 8 32; do timeout 120 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/${TAG}_round_budget_B$b.json 2>&1; done
cp gpurun_out/${TAG}_round_budget_B32.json profiles/${TAG}_round_budget_B32.json
timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/${TAG}_round_gaps_B32.txt 2>&1
timeout 120 python scripts/r04/round_gaps.py 1 3000 240 > gpurun_out/${TAG}_round_gaps_B1.txt 2>&1
FRX_SETUP_TIMING=1 timeout 500 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_headline.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/bench.err | cut -c1-200
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form.json 2> gpurun_out/bench_d.err; echo "bench driver form rc=$?"
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $(find $R/gpurun_out/prof_final -name "fin_kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_headline.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-160 $R/gpurun_out/${TAG}_kernel_stats_headline.csv | head -9
cd $R
python - <<PY
import json
for f in ('gpurun_out/${TAG}_bench_headline.json', 'gpurun_out/${TAG}_bench_driver_form.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    keys = ['value','ms_per_step','ms_per_step_direct_launches','ms_per_step_host_wall','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','plan_setup_ms','plan_initial_guess_ms','plan_ms_with_setup',
            'plan_coeff_spread_vs_cpu','plan_coeff_spread_cpu_vs_cpu','plan_objective_spread_vs_cpu','plan_objective_spread_cpu_vs_cpu','plan_ms_per_stage_path','plan_resident_failed']
    print(f, {k: d.get(k) for k in keys}); print(r['stage_kernels_us'], 'frac', r['frac'], 'large', r['large_batch'], 'valu', r['valu'], 'traffic', r['traffic'], 'fp64', {k: v for k, v in (r['fp64'] or {}).items() if k != 'dynamic'})
    print('round', {k: v for k, v in (r['round'] or {}).items() if k != 'budget'})
    print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k not in ('sample', 'note')})
PY
for c in plumbing synthetic8; do timeout 300 python bench.py --config $c --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/bench_$c.err; echo "$c rc=$?"; done
timeout 400 python bench.py --config montecarlo4096 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_montecarlo4096.json 2> gpurun_out/bench_mc.err; echo "mc rc=$?"
python - <<PY
import json
for c in ('plumbing','synthetic8','montecarlo4096'):
    try:
        d=json.loads(open('gpurun_out/${TAG}_bench_%s.json' % c).read().strip().splitlines()[-1]); r=d['roofline']
        print(c, {k: (round(d[k], 3) if isinstance(d.get(k), float) else d.get(k)) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_per_stage_path','plan_ms_with_setup','plan_path','plans_per_s','plans_per_s_per_stage_path','plans_per_s_work_queue','plan_taken_over','work_queue_verdict_mismatches']}, r['stage_kernels_us'], 'frac', r['frac'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('plan_ms_one_candidate_1thread'))
    except Exception as e: print(c, 'failed', e)
PY
FRX_BENCH_DEVICE=0 FRX_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline 2> gpurun_out/bench_2ranks.err | tail -1 > gpurun_out/${TAG}_bench_2ranks_self_launched_one_device.json
FRX_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --multi lib --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline 2> gpurun_out/bench_lib.err | tail -1 > gpurun_out/${TAG}_bench_2shards_lib_one_device.json
cp gpurun_out/bench_8ranks_one_device.json gpurun_out/${TAG}_bench_8ranks_one_device.json 2>/dev/null; cp gpurun_out/host_budget_8_ranks.json gpurun_out/${TAG}_host_budget_8_ranks.json 2>/dev/null
python -c "
import json
for f in ('gpurun_out/${TAG}_bench_2ranks_self_launched_one_device.json','gpurun_out/${TAG}_bench_2shards_lib_one_device.json'):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','n_gpus','plan_ms','winner_id','plan_status_ok']}, d['config']['front_end'])
    except Exception as e: print(f, 'failed', e)"
