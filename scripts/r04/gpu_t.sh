cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider --timeout 300 2>&1 | tail -5
