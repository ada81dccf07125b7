cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -s -k "independent or lockstep" 2>&1 | grep -E "cand|knot_pcr:|banded_lu:|passed|failed"
