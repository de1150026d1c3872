cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decode_roles.py -x -q > $O/pytest_roles.log 2>&1; echo "pytest rc=$?" >> $O/pytest_roles.log
tail -25 $O/pytest_roles.log
