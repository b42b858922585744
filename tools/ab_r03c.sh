#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() {
  local label=$1; shift
  env "$@" python3 bench.py --no-extras --cpu-seconds 0 --steps 80 --warmup 20 2>gpurun_out/err.log | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$label', 'ms/step', round(d['ms_per_step'],4), 'l1fwd+reduce us', round(d['roofline']['avg_us'],1), 'score us', round(d['roofline_score']['avg_us'],1), 'loss', round(d['config'].get('final_loss',0),2))" || tail -5 gpurun_out/err.log
}
for rep in 1 2 3; do
run natural X=1


done
