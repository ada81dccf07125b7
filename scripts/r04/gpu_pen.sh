cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in 1 2 4 8; do echo "== groups per workgroup $g"; FRX_PENALTY_GROUPS=$g timeout 200 python scripts/r04/penalty_groups.py 2>&1 | grep -v "^$" | tee -a gpurun_out/r04_penalty_groups.jsonl | cut -c1-200; done
