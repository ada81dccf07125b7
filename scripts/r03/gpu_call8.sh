cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 32 8 1; do FRX_RESIDENT_HOST_STATS=1 timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/r03_c8_budget_B$b.json 2> gpurun_out/r03_c8_hoststats_B$b.txt; python - <<PY
import json
s=open('gpurun_out/r03_c8_budget_B$b.json').read(); d=json.loads(s[:s.rindex('}\n{')+1]) if '}\n{' in s else json.loads(s)
print($b, d['leader'], [v for k,v in d.items() if k.startswith('host_wait')])
PY
grep "mailbox thread" gpurun_out/r03_c8_hoststats_B$b.txt | tail -4; done
