cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c4; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_decode_roles.py -x -q 2>&1 | tail -2
for c in 256 128; do TASU_ROLES_CUS=$c timeout 300 python tools/bench_paths.py decode 16 2>&1 | grep '"what"' | sed "s/^/cus=$c /" | cut -c1-150; done | tee $O/decode_ab.log
