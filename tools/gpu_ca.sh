cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -k "cross or softmax or ca_" 2>&1 | tail -6
