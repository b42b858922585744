#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_abl.so
for d in 0 3 7 15 23 31 63; do echo "DBG=$d: $(SNAG_GEMM_DBG=$d python3 tools/probe_small_gemm.py 2>&1 | tail -1)"; done
