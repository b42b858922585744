#!/bin/bash
cd "$GRAFT_REPO_ROOT"
p() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],4), round(d['value']))"; }
for i in 1 2 3; do
  for e in "X=1" "SNAG_GEMM_TALLK=-1" "SNAG_GEMM_TALLK=2048" "SNAG_GEMM_TALLK=512 SNAG_GEMM_TARGET=512"; do
  env $e python3 bench.py --no-extras --cpu-seconds 0 --steps 100 2>/dev/null | p "[$e]"
  done
done
SNAG_GEMM_LOG=1 python3 bench.py --no-extras --cpu-seconds 0 --steps 1 --warmup 0 2>&1 | grep "snag_gemm" | sort | uniq -c | sort -rn | head -30
