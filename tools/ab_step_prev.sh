#!/bin/bash
# same-box A/B of the whole step: the working tree (library + engine) against another commit checked out and built under gpurun_prev/:
#   rm -rf gpurun_prev; mkdir gpurun_prev; git archive <commit> | tar -x -C gpurun_prev; (cd gpurun_prev && python -m snag_amd.build); tools/ab_step_prev.sh   (on the GPU box)
cd "$GRAFT_REPO_ROOT"
run() {
  ( cd "$1" && python3 bench.py --no-extras --cpu-seconds 0 --steps 80 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$2', 'ms/step', round(d['ms_per_step'],4))" )
}
for rep in 1 2 3; do run "$GRAFT_REPO_ROOT" tree; run "$GRAFT_REPO_ROOT/gpurun_prev" prev; done
