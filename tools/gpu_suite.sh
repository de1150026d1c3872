cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
