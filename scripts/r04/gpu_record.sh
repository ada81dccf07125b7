# round 4 record call: parity suite, counter passes, bench line, rocprofv3 kernel statistics of the same command, round budgets and timeline statistics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r04}
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 200 --durations=5 > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/r04/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log | cut -c1-300
cd $GRAFT_REPO_ROOT
cp gpurun_out/r04_pmc_headline.json profiles/r04_pmc_headline.json            # the bench line below reads the counters of THIS call ("from_profile")
for b in 1 8 32; do timeout 120 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/${TAG}_round_budget_B$b.json 2>&1; done
cp gpurun_out/${TAG}_round_budget_B32.json profiles/${TAG}_round_budget_B32.json   # (roofline.round.budget of the bench line: this call's)
timeout 120 python scripts/r04/round_gaps.py 32 3000 240 > gpurun_out/${TAG}_round_gaps_B32.txt 2>&1
timeout 120 python scripts/r04/round_gaps.py 1 3000 240 > gpurun_out/${TAG}_round_gaps_B1.txt 2>&1
FRX_SETUP_TIMING=1 timeout 500 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_headline.json 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "frx setup" gpurun_out/bench.err | tail -6; tail -1 gpurun_out/bench.err
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $(find $R/gpurun_out/prof_final -name "fin_kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_headline.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-160 $R/gpurun_out/${TAG}_kernel_stats_headline.csv | head -9
cd $R
python - <<PY
import json
d = json.loads(open('gpurun_out/${TAG}_bench_headline.json').read().strip().splitlines()[-1]); r = d['roofline']
keys = ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','plan_setup_ms','plan_initial_guess_ms','plan_ms_with_setup','plan_ms_with_setup_one_candidate',
        'plan_coeff_spread_vs_cpu','plan_coeff_spread_cpu_vs_cpu','plan_objective_spread_vs_cpu','plan_objective_spread_cpu_vs_cpu','plan_ms_per_stage_path','plan_resident_failed']
print({k: d.get(k) for k in keys}); print(r['stage_kernels_us'], 'frac', r['frac'], 'large', r['large_batch'], 'valu', r['valu'], 'traffic', r['traffic'])
print('round', r['round'])
print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k != 'sample'})
PY
timeout 300 python bench.py --config montecarlo4096 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_montecarlo4096.json 2> gpurun_out/bench_mc.err
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench_montecarlo4096.json').read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('plan') or k in ('value','work_queue_equals_default_path_status')})"
FRX_BENCH_DEVICE=0 FRX_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline 2> gpurun_out/bench_2ranks.err | tail -1 > gpurun_out/${TAG}_bench_2ranks_self_launched_one_device.json
FRX_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --multi lib --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline 2> gpurun_out/bench_lib.err | tail -1 > gpurun_out/${TAG}_bench_2shards_lib_one_device.json
python -c "
import json
for f in ('gpurun_out/${TAG}_bench_2ranks_self_launched_one_device.json','gpurun_out/${TAG}_bench_2shards_lib_one_device.json'):
    d=json.loads(open(f).read()); print(f.split('/')[-1], {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','n_gpus','plan_ms','winner_id','plan_status_ok']}, d['config']['front_end'])"
