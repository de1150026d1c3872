cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or rope" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
tail -5 $O/pytest_attn.log
timeout 600 python tools/bench_attn_gqa.py > $O/bench_attn.log 2>&1; cat $O/bench_attn.log | tail -8
