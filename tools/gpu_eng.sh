cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_lora.py -x -q -k "rank_above" 2>&1 | tail -6
