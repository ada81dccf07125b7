cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 900 python scripts/kernel_sweep.py --full --states it60 2>&1 | grep -v amdgpu.ids | grep -v Traceback -A0 | grep "^{"
