cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ht in 4 8 16 32; do
  for fw in 0 1; do
    echo "host threads $ht forward $fw: $(FRX_RESIDENT_HOST_THREADS=$ht FRX_RESIDENT_FORWARD=$fw timeout 100 python scripts/r03/plan_once.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['B=32']['us_per_round'], d['B=1']['us_per_round'])")"
  done
done
FRX_RESIDENT_HOST_THREADS=16 FRX_RESIDENT_FORWARD=0 timeout 120 python scripts/r04/round_gaps.py 32 3000 240 2>&1 | grep -E "rounds_in|adj_end|confirmed"
FRX_RESIDENT_HOST_STATS=1 FRX_RESIDENT_FORWARD=0 timeout 100 python scripts/r03/plan_once.py 2>&1 | grep "mailbox thread" | head -8
nproc; cat /proc/loadavg
