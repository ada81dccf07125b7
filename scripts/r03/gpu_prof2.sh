cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 1 8 32; do FRX_PROFILE_MODE=2 timeout 300 python scripts/resident_profile.py $b 64 16 3000 > gpurun_out/p2_B$b.json 2>&1; python - <<PY
import json
t=open('gpurun_out/p2_B$b.json').read()
d=json.loads(t[:t.index('\n}\n')+2])
print($b, d['us_per_round_wall'], d['leader'])
PY
done
