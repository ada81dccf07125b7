# round 3 record call: parity suite, counter passes, bench line, rocprofv3 kernel statistics of the same command, round budgets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r03}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 --durations=5 > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests.log | head -20; grep -A6 "slowest" gpurun_out/tests.log | tail -6
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/r03/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/r03_pmc_headline.json profiles/r03_pmc_headline.json            # the bench line below reads the counters of THIS call ("from_profile")
timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_headline.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.err
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $(find $R/gpurun_out/prof_final -name "fin_kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_headline.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-160 $R/gpurun_out/${TAG}_kernel_stats_headline.csv | head -9
cd $R
for b in 1 8 32; do timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/${TAG}_round_budget_B$b.json 2>&1; done
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_headline.json')); r=d['roofline']; print({k:d[k] for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_ms_per_stage_path','plan_resident_failed']}, r['stage_kernels_us'], 'frac', r['frac'], 'eval', r['evaluation']['frac'], 'large', r['large_batch'], 'valu', r['valu'], 'traffic', r['traffic'], 'knot', {k:v['traffic_range'] for k,v in r['knot_kernels'].items()}, 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['plan_ms_batch'])"
head -30 gpurun_out/${TAG}_round_budget_B32.json | tr -d '\n '; echo
timeout 900 python scripts/r03/rccl_init_probe.py 2>&1 | tail -6
