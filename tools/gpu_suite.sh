set -euo pipefail; cd "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"; export TMPDIR=/tmp; set +e   # (the runs below report their own exit codes)
O=gpurun_out/suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
