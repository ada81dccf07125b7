# round 3 record call: parity suite, counter passes, bench line, rocprofv3 kernel statistics of the same command, round budgets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r03}
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 200 --durations=5 > gpurun_out/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests.log | head -20; grep -A6 "slowest" gpurun_out/tests.log | tail -6
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/r03/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/r03_pmc_headline.json profiles/r03_pmc_headline.json            # the bench line below reads the counters of THIS call ("from_profile")
timeout 400 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_headline.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.err
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $(find $R/gpurun_out/prof_final -name "fin_kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_headline.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-160 $R/gpurun_out/${TAG}_kernel_stats_headline.csv | head -9
cd $R
for b in 1 8 32; do timeout 120 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/${TAG}_round_budget_B$b.json 2>&1; done
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_headline.json')); r=d['roofline']; print({k:d[k] for k in ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_ms_per_stage_path','plan_resident_failed']}, r['stage_kernels_us'], 'frac', r['frac'], 'eval', r['evaluation']['frac'], 'large', r['large_batch'], 'valu', r['valu'], 'traffic', r['traffic'], 'knot', {k:v['traffic_range'] for k,v in r['knot_kernels'].items()}, 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['plan_ms_batch'])"
head -30 gpurun_out/${TAG}_round_budget_B32.json | tr -d '\n '; echo
# the work queue against the per-stage path over batch sizes; the Monte-Carlo share (512 scenarios per GPU) with both paths on the bench line;
# two ranks on one device (N > 1 control flow of bench.py: cpu_baseline and roofline on rank 0)
rm -f gpurun_out/queue_sizes.jsonl
timeout 200 python scripts/r03/queue_sizes.py 33 64 96 128 512 > gpurun_out/queue_sizes.log 2>&1; tail -2 gpurun_out/queue_sizes.log | cut -c1-400
timeout 300 python bench.py --config montecarlo4096 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_montecarlo4096.json 2> gpurun_out/bench_mc.err
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench_montecarlo4096.json').read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('plan') or k in ('value','work_queue_equals_default_path_status')})"
FRX_BENCH_DEVICE=0 FRX_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 50 --warmup 10 --large-batch 0 2> gpurun_out/bench_2ranks.err | tail -1 > gpurun_out/${TAG}_bench_2ranks_one_device.json
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench_2ranks_one_device.json').read()); print('2 ranks on one device', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','n_gpus','plan_ms','plan_path','winner_id','winner_rank','plan_status_ok']}, 'cpu_baseline' , d['cpu_baseline'] is not None, 'roofline', d['roofline']['frac'])"
timeout 120 python scripts/r03/rccl_init_probe.py 2>&1 | tail -3
