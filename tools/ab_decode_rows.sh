#!/bin/bash
# A/B of the rows-per-workgroup of the decode step's row-wise kernels (norm, slab finish): ms per position, separate processes.
# The switches exist in the LAB build only: make -C ps_slm_amd/csrc lab; export TASU_LIB_PATH=ps_slm_amd/libtasu_hip_lab.so
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
for rep in 1 2; do
  for r in 4 1; do
    echo "rows=$r: $(TASU_FINISH_ROWS=$r TASU_NORM_ROWS=$r python tools/bench_paths.py decode 16 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/position")')"
  done
done
for r in 4 1; do
  echo "7B rows=$r: $(TASU_FINISH_ROWS=$r TASU_NORM_ROWS=$r python tools/bench_paths.py decode 16 qwen2.5-7b 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/position")')"
done
