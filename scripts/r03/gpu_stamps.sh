cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 1 32; do for r in 1500 1501 1502; do FRX_RESIDENT_STAMP_ROUND=$r timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/st_B$b.json 2>&1; python - <<PY
import json
t=open('gpurun_out/st_B$b.json').read()
d=json.loads(t[:t.index('\n}\n')+2])
print($b, $r, d['leader']['backward'], d['forward_stamps'], d['adjoint_stamps'])
PY
done; done
