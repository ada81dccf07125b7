# round 6, third call: kernel arguments by pointer (k_eval_cluster, k_round) against by value, alternating processes on one box; parity of the touched paths first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py tests/test_takeover.py -m gpu -q -p no:cacheprovider --timeout 500 > gpurun_out/r06_tests3.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r06_tests3.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r06_tests3.log | head -30
AB_KAPPA48=1 timeout 1500 python scripts/ab_env.py "FRX_ROUND_ARGPTR=0 FRX_EVAL_ARGPTR=0" "-" 4 > gpurun_out/r06_ab_argptr.jsonl 2> gpurun_out/ab_argptr.err; echo "ab rc=$?"; tail -1 gpurun_out/r06_ab_argptr.jsonl
