cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "device_vector" 2>&1 | grep -E "^E|assert|Error" | head -12
