# end-of-round artefacts (round 2): parity suite, counter passes, bench line, rocprofv3 kernel statistics of the same command, round budgets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/tests.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_pmc_round2.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/r02_pmc_headline.json profiles/r02_pmc_headline.json            # the bench line below reads the counters of THIS call
timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.err
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $R/gpurun_out/prof_final/fin_kernel_stats.csv $R/gpurun_out/kernel_stats_final.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-160 $R/gpurun_out/kernel_stats_final.csv | head -9
cd $R
for b in 1 8 32; do timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/round_budget_B$b.json 2>&1; done
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print({k:d[k] for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_ms_per_stage_path']}, d['roofline']['stage_kernels_us'], d['roofline']['frac'], d['roofline']['penalty']['large_batch'], d['roofline']['penalty']['valu'], d['roofline']['penalty']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['plan_ms_batch'])"
