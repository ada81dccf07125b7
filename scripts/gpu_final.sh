# end-of-round artefacts: bench line + rocprofv3 kernel statistics of the same command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.err
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o fin -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_final.err
cp $R/gpurun_out/prof_final/fin_kernel_stats.csv $R/gpurun_out/kernel_stats_final.csv; rm -rf $R/gpurun_out/prof_final
cut -c1-160 $R/gpurun_out/kernel_stats_final.csv | head -8
python -c "
import json; d=json.load(open('$R/gpurun_out/bench.json')); print({k:d[k] for k in ['value','ms_per_step','plan_ms','plan_rounds']}, d['roofline']['avg_kernel_us'], d['roofline']['frac'], d['roofline']['large_batch']['frac'])"
