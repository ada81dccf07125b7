cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 1 4 8 16 32; do FRX_RESIDENT_TIMED_READ=1 timeout 300 python scripts/resident_profile.py $b 64 16 1500 > gpurun_out/pcie_B$b.json 2>&1; python - <<PY
import json
t=open('gpurun_out/pcie_B$b.json').read()
d=json.loads(t[:t.index('\n}\n')+2])
print($b, d['us_per_round_wall'], d.get('timed_host_read_us'), d['leader']['backward'], d['leader']['post'])
PY
done
