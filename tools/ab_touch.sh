#!/bin/bash
# A/B of the first-touch variants of the streaming decode GEMMs (make -C ps_slm_amd/csrc touch MASK=n): ms per position, separate processes
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
for rep in 1 2; do
  for lib in libtasu_hip.so libtasu_hip_touch4.so libtasu_hip_touch20.so libtasu_hip_touch2.so libtasu_hip_touch31.so; do
    [ -f ps_slm_amd/$lib ] || continue
    echo "$lib: $(TASU_LIB_PATH=$PWD/ps_slm_amd/$lib python tools/bench_paths.py decode 16 2>/dev/null | cut -c100-190)"
  done
done
