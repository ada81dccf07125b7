# round 3, call 4: early command read behind the adjoint's first loads; host service thread sweep; budgets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/r03/multi_probe.py > gpurun_out/multi_probe.txt 2>&1; grep -E "scenario|single|per-stage" gpurun_out/multi_probe.txt
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > gpurun_out/tests_c4.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/tests_c4.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests_c4.log | head -40
for b in 32 1; do timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/r03_c4_budget_B$b.json 2>&1; head -42 gpurun_out/r03_c4_budget_B$b.json | tr -d '\n '; echo; done
timeout 600 python scripts/r03/host_threads.py 2>&1 | tee gpurun_out/r03_host_threads.jsonl | tail -20
