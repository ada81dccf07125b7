cd $GRAFT_REPO_ROOT
timeout 900 python scripts/r03/ab_libs.py ab_base . 6 32 | python -c "
import sys,json,numpy as np
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items(): a=np.array([x['us_per_round'] for x in v]); print(k.split('/')[-1], 'rounds',v[0]['rounds'],'per-process medians',np.round(np.median(a,axis=1),2),'median',round(float(np.median(a)),2),'min',round(float(a.min()),2))
"
