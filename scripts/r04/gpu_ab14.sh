cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python scripts/r03/ab_libs.py ab_late . 3 32 | tee gpurun_out/r04_ab14_B32.json
timeout 300 python scripts/r03/ab_libs.py ab_late . 2 1 | tee gpurun_out/r04_ab14_B1.json
timeout 120 python scripts/r04/round_timeline.py 32 3000 3 2>&1 | grep dense | head -12
cp -r ab_late/fast-racing_amd/libfrx.so fast-racing_amd/libfrx.so
timeout 120 python scripts/r04/round_timeline.py 32 3000 3 2>&1 | grep dense | head -12
