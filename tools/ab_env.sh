#!/bin/bash
# tools/ab_env.sh <variant|base> "ENV=.. ENV=.." ["ENV=.."...]  -> probe_gemm_step under each environment
cd "$GRAFT_REPO_ROOT"
v=$1; shift
if [ "$v" != base ]; then export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
for e in "$@"; do
  echo "== $v [$e]"
  env $e python3 tools/probe_gemm_step.py 2>&1 | grep -v amdgpu.ids
done
