#!/bin/bash
# round-3 A/B of the dominant projection GEMM inside the step: tools/ab_r03.sh  (variants under snag_amd/variants/)
cd "$GRAFT_REPO_ROOT"
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-extras --cpu-seconds 0 --steps 80 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$label', 'ms/step', round(d['ms_per_step'],4), 'l1fwd+reduce us', round(d['roofline']['avg_us'],1), 'score us', round(d['roofline_score']['avg_us'],1), 'loss', round(d['config'].get('final_loss',0),2))"
}
for rep in 1 2; do
run base X=1
run dbg1_noA SNAG_GEMM_DBG=1
run dbg2_noB SNAG_GEMM_DBG=2
run dbg3_none SNAG_GEMM_DBG=3
run l1kc SNAG_L1_KC=1
for v in "$@"; do run "var_$v" SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; run "var_${v}_l1kc" SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so SNAG_L1_KC=1; done
done
