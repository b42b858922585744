#!/bin/bash
# in-step A/B of the working-tree library against variants: tools/ab_pair.sh NAME [NAME...]   (variants under snag_amd/variants/)
cd "$GRAFT_REPO_ROOT"
run() {
  local label=$1; shift
  env "$@" python3 bench.py --no-extras --cpu-seconds 0 --steps 80 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$label', 'ms/step', round(d['ms_per_step'],4), 'l1fwd+reduce us', round(d['roofline']['avg_us'],1), 'score us', round(d['roofline_score']['avg_us'],1))"
}
for rep in 1 2 3; do
run tree X=1
for v in "$@"; do run "var_$v" SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; done
done
