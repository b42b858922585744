#!/bin/bash
cd "$GRAFT_REPO_ROOT"
p() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],4), round(d['value']))"; }
V=$PWD/snag_amd/variants/libsnag_hip_${1:-sched}.so
for i in 1 2 3; do
  python3 bench.py --no-extras --cpu-seconds 0 --steps 100 2>/dev/null | p "base"
  SNAG_HIP_LIB=$V python3 bench.py --no-extras --cpu-seconds 0 --steps 100 2>/dev/null | p "variant"
done
