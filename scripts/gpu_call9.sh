# f4 e2e + full GPU suite + end-of-round artefacts (bench line, rocprofv3 kernel statistics, round budgets)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export FRX_ROUND_TIMEOUT_MS=3000
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.txt
tail -3 gpurun_out/gpu_tests.txt; grep -h "^route\|cells," gpurun_out/gpu_tests.txt | head
unset FRX_ROUND_TIMEOUT_MS
timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.err
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $R/gpurun_out/prof_final/fin_kernel_stats.csv $R/gpurun_out/kernel_stats_final.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-160 $R/gpurun_out/kernel_stats_final.csv | head -8
cd $R
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print({k:d.get(k) for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_ms_per_stage_path','plan_ms_one_candidate']}, d['roofline'].get('avg_kernel_us'), d['roofline']['frac'])"
timeout 300 python scripts/resident_profile.py 1 64 16 3000 > gpurun_out/round_budget_B1.json 2>&1; tail -3 gpurun_out/round_budget_B1.json | cut -c1-200
timeout 300 python scripts/resident_profile.py 8 64 16 3000 > gpurun_out/round_budget_B8.json 2>&1; tail -3 gpurun_out/round_budget_B8.json | cut -c1-200
