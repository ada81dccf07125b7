# build against build, many alternating processes (noisy boxes): bash scripts/r04/gpu_ab_many.sh [variant] [reps] [B]
cd $GRAFT_REPO_ROOT
VAR=${1:-base}; REPS=${2:-6}; BB=${3:-32}
timeout 300 python -m pytest tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 300 2>&1 | tail -1
timeout 900 python scripts/r03/ab_libs.py ab_$VAR ${4:-.} $REPS $BB | python -c "
import sys,json,numpy as np
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items(): a=np.array([x['us_per_round'] for x in v]); print(k.split('/')[-1], 'rounds',v[0]['rounds'],'per-process medians',np.round(np.median(a,axis=1),2),'median',round(float(np.median(a)),2),'min',round(float(a.min()),2))
"
