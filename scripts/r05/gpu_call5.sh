# round 5, call 5: the forward map with its per-candidate constants handed in by the leader (working tree) against the tree before (ab_head) and round 4 (ab_r04)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in ab_head .; do
FRX_ROOT=$v timeout 200 python scripts/resident_profile.py 32 64 16 3000 2>&1 | python -c "
import sys,json
t=sys.stdin.read().split('\n{\"per_stage')[0]
d=json.loads(t)
print('$v', json.dumps({k:d[k] for k in ('us_per_round_wall','forward_stamps')}))"
done
timeout 1200 python scripts/r05/ab_all.py ab_r04 ab_head . 5 > gpurun_out/ab5.jsonl 2> gpurun_out/ab5.err; tail -3 gpurun_out/ab5.jsonl | cut -c1-700
