cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_raw_features.py tests/test_gpu_model.py -x -q 2>&1 | tail -12
