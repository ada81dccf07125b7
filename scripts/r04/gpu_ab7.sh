cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_resident.py tests/test_gpu_multi.py -m gpu -q -x -p no:cacheprovider --timeout 120 > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/ab_tests.log | tail -3
for numa in 0 1; do for fw in 0 1; do
  echo "numa $numa forward $fw: $(FRX_NUMA=$numa FRX_RESIDENT_FORWARD=$fw timeout 100 python scripts/r03/plan_once.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['B=32']['us_per_round'], d['B=1']['us_per_round'])")"
done; done
for numa in 0 1; do for fw in 0 1; do
  echo "numa $numa forward $fw: $(FRX_NUMA=$numa FRX_RESIDENT_FORWARD=$fw timeout 100 python scripts/r03/plan_once.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['B=32']['us_per_round'], d['B=1']['us_per_round'])")"
done; done
echo "gaps numa 1 forward 0: $(FRX_RESIDENT_FORWARD=0 timeout 120 python scripts/r04/round_gaps.py 32 3000 240 2>&1 | grep -E 'adj_end' | tr '\n' '|')"
echo "gaps numa 1 forward 1: $(FRX_RESIDENT_FORWARD=1 timeout 120 python scripts/r04/round_gaps.py 32 3000 240 2>&1 | grep -E 'adj_end' | tr '\n' '|')"
