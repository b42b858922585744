#!/bin/bash
# bench.py under several libraries, alternating on one box; prints ms/step and one record of the line:  tools/ab_bench_field.sh FIELD NAME...  ("base" = in-tree)
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
F=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
  python3 bench.py --steps 60 --warmup 15 --cpu-seconds ${AB_CPU:-0} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', 'ms/step', round(d['ms_per_step'],4), json.dumps(d.get('$F')))"
done
