set -x
cd $GRAFT_REPO_ROOT
for pad in 0 64 128 32 192 1088; do
  echo "=== pad $pad"
  timeout 300 python tools/bench_gemm.py --cold --pad $pad --only qkv,o,gate_up,down,d_down,d_gate_up,d_qkv,sq4096,sq8192 2>&1 | grep -v '^\[' 
done
