# final state after the last kernel change: parity suite, smoke, bench line, kernel statistics of the same command, round budgets
# (counter passes, queue sizes, Monte-Carlo and two-rank lines: scripts/r03/gpu_record.sh, same day, kernels they measure unchanged)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r03}
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 150 > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -2; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests.log | head -10
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_headline.json 2> gpurun_out/bench.err
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $(find $R/gpurun_out/prof_final -name "fin_kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_headline.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-140 $R/gpurun_out/${TAG}_kernel_stats_headline.csv | head -6
cd $R
for b in 32 1; do FRX_PROFILE_MODE=2 timeout 100 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/${TAG}_round_budget_B$b.json 2>&1; done
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_headline.json')); r=d['roofline']; print({k:d[k] for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_ms_per_stage_path','plan_resident_failed']}, r['stage_kernels_us'], 'frac', r['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['plan_ms_batch'])"
head -24 gpurun_out/${TAG}_round_budget_B32.json | tr -d '\n '; echo
