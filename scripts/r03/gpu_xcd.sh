cd $GRAFT_REPO_ROOT
for b in 8 32; do FRX_RESIDENT_HOST_STATS=1 timeout 300 python scripts/resident_profile.py $b 64 16 300 2>&1 | grep "clusters on one XCD" | tail -1 | cut -c1-600; done
