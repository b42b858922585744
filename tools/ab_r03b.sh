#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() {
  local label=$1; shift
  env "$@" python3 bench.py --no-extras --cpu-seconds 0 --steps 80 --warmup 20 2>gpurun_out/err.log | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$label', 'ms/step', round(d['ms_per_step'],4), 'l1fwd+reduce us', round(d['roofline']['avg_us'],1), 'score us', round(d['roofline_score']['avg_us'],1), 'loss', round(d['config'].get('final_loss',0),2))" || tail -5 gpurun_out/err.log
}
for rep in 1 2; do
run base X=1
run early SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_early.so
run base_dbg3 SNAG_GEMM_DBG=3
run ldt+32 SNAG_LDT_EXTRA=32
run ldt+352 SNAG_LDT_EXTRA=352
run ldt+992 SNAG_LDT_EXTRA=992
done
