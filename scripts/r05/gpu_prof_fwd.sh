cd $GRAFT_REPO_ROOT
for v in . ab_fwd; do
FRX_ROOT=$v timeout 200 python scripts/resident_profile.py 32 64 16 3000 2>&1 | python -c "
import sys,json
t=sys.stdin.read().split('\n{\"per_stage')[0]
d=json.loads(t)
print('$v', json.dumps({k:d[k] for k in ('us_per_round_wall','forward_stamps','adjoint_stamps')}))"
done
