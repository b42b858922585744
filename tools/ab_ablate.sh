#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_abl.so
for t in "SNAG_GEMM_TALLK=1" "X=1"; do
for d in 0 3 7 15 23 31 63 39; do
  echo "== $t DBG=$d: $(env $t SNAG_GEMM_DBG=$d python3 tools/probe_gemm_step.py 2>&1 | grep 'fd=4096')"
done
done
