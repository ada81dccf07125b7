# round 4, call 2: the bench line with the new fields (set-up timing on stderr) and both N > 1 front ends of bench.py on this 1-GPU box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
FRX_SETUP_TIMING=1 timeout 500 python bench.py --steps 200 --warmup 20 > gpurun_out/r04_c1_bench_headline.json 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "frx setup" gpurun_out/bench.err | head -12; tail -2 gpurun_out/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_c1_bench_headline.json').read().strip().splitlines()[-1]); r = d['roofline']
keys = ['value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','plan_setup_ms','plan_initial_guess_ms','plan_ms_with_setup',
        'plan_setup_ms_one_candidate','plan_initial_guess_ms_one_candidate','plan_ms_with_setup_one_candidate','plan_coeff_spread_vs_cpu','plan_coeff_spread_cpu_vs_cpu','plan_objective_spread_vs_cpu','plan_objective_spread_cpu_vs_cpu','plan_resident_failed']
print({k: d.get(k) for k in keys}); print(r['stage_kernels_us'], 'frac', r['frac'], 'round', {k: r['round'][k] for k in ('us_per_round','frac','fp64_frac')} if r.get('round') else None)
print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k != 'sample'})
PY
FRX_BENCH_DEVICE=0 FRX_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline 2> gpurun_out/bench_2ranks.err | tail -1 > gpurun_out/r04_c1_bench_2ranks_self_launched.json; echo "2 ranks rc=$?"; tail -2 gpurun_out/bench_2ranks.err
python -c "
import json; d=json.loads(open('gpurun_out/r04_c1_bench_2ranks_self_launched.json').read()); print('self-launched', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','n_gpus','plan_ms','winner_id','winner_rank','plan_status_ok']}, d['config']['front_end'])"
FRX_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --multi lib --steps 50 --warmup 10 --large-batch 0 --no-cpu-baseline 2> gpurun_out/bench_lib.err | tail -1 > gpurun_out/r04_c1_bench_2shards_lib.json; echo "lib rc=$?"; tail -2 gpurun_out/bench_lib.err
python -c "
import json; d=json.loads(open('gpurun_out/r04_c1_bench_2shards_lib.json').read()); print('lib', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ['value','n_gpus','plan_ms','plan_ms_whole_job','plan_shards','plan_winner_exchange','winner_id','lib_winner_id','plan_status_ok_whole_job']}, d['config']['front_end'])"
