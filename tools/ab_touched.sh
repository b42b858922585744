#!/bin/bash
# tools/probe_touched.py under several libraries: tools/ab_touched.sh NAME...   ("base" = in-tree)
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
for v in "$@"; do
  if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
  python3 tools/probe_touched.py 2>&1 | grep "B="
done
