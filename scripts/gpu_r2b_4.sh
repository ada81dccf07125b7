cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 60 scripts/micro/mailbox_probe 32 2000 3000 2>&1 | tee gpurun_out/mailbox_probe.txt
timeout 60 scripts/micro/mailbox_probe 1 2000 3000 2>&1 | tee -a gpurun_out/mailbox_probe.txt
timeout 300 python scripts/resident_profile.py 1 64 16 3000 > gpurun_out/budget_B1.json 2>&1; head -62 gpurun_out/budget_B1.json | tr -d '\n ' ; echo
