#!/bin/bash
# ablation of the eight-wave ping-pong loop inside the step (needs the abl variant: tools/build_variant.sh abl -DSNAG_GEMM_ABLATE=1)
cd "$GRAFT_REPO_ROOT"
export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_abl.so
run() {
  local label=$1; shift
  env "$@" python3 bench.py --no-extras --cpu-seconds 0 --steps 80 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$label', 'ms/step', round(d['ms_per_step'],4), 'l1fwd+reduce us', round(d['roofline']['avg_us'],1))"
}
for rep in 1 2; do
run base X=1
run noMFMA SNAG_GEMM_DBG=8
run noAsplit SNAG_GEMM_DBG=16
run noBsplit SNAG_GEMM_DBG=64
run noSplit SNAG_GEMM_DBG=80
run noLoads SNAG_GEMM_DBG=35
run noSplit_noLoads SNAG_GEMM_DBG=115
run onlyMFMA_frag SNAG_GEMM_DBG=115
run noMFMA_noLoads SNAG_GEMM_DBG=43
done
