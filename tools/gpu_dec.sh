cd $GRAFT_REPO_ROOT
python tools/ab_decode_split.py 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q -k "decode or generate or stream or skinny or slab" 2>&1 | tail -5
