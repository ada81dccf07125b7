# round 2, call 1: hand-off probe for the persistent round kernel, the whole -m gpu suite (new: configs[3]/[4] shares, delta curve,
# fault injection), and a baseline bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 scripts/micro/cluster_probe > gpurun_out/probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/probe.txt
timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.txt
timeout 400 python bench.py > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; echo "bench rc=$?" >> gpurun_out/bench_r2a.err
tail -5 gpurun_out/probe.txt; tail -15 gpurun_out/gpu_tests.txt; tail -c 600 gpurun_out/bench_r2a.json
