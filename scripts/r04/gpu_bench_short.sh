# one short headline bench line (no CPU baseline, no large-batch leg): the fields a box-to-box comparison needs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch 0 2>/dev/null | python -c "
import sys,json,socket; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'host': socket.gethostname(), **{k:d[k] for k in ('value','ms_per_step','plan_ms','plan_rounds','plan_us_per_round','plan_ms_one_candidate','plan_us_per_round_one_candidate','plan_setup_ms','plan_setup_ms_first_handle_of_this_kind_in_the_process','plan_initial_guess_ms')}}))" | tee -a gpurun_out/r04_bench_boxes.jsonl
