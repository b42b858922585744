#!/bin/bash
cd "$GRAFT_REPO_ROOT"
p() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],4), round(d['value']))"; }
for i in 1 2 3; do
  for e in "X=1" "SNAG_GEMM_TALLK=1024"; do
  env $e python3 bench.py --no-extras --cpu-seconds 0 --steps 100 2>/dev/null | p "[$e]"
  done
done
