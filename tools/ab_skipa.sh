#!/bin/bash
# timing experiment: the dominant GEMM loop with the A split's VALU removed (wrong products), stamps + in-step time
cd "$GRAFT_REPO_ROOT"
V=$PWD/snag_amd/variants
for kc in 0 1; do
echo "== stamps, STAMP_KC=$kc, production split"; SNAG_HIP_LIB=$V/libsnag_hip_stamp.so STAMP_KC=$kc python3 tools/stamp_gemm.py 2>/dev/null | grep -v amdgpu.ids
echo "== stamps, STAMP_KC=$kc, A split skipped"; SNAG_HIP_LIB=$V/libsnag_hip_stampskip.so STAMP_KC=$kc python3 tools/stamp_gemm.py 2>/dev/null | grep -v amdgpu.ids
done
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-extras --cpu-seconds 0 --steps 80 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$label', 'ms/step', round(d['ms_per_step'],4), 'l1fwd+reduce us', round(d['roofline']['avg_us'],1))"
}
for rep in 1 2; do
run base X=1
run skipa SNAG_HIP_LIB=$V/libsnag_hip_skipa.so
done
