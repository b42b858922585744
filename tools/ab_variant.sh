#!/bin/bash
# A/B library variants on one device: tools/ab_variant.sh <name> [<name> ...]   (names under snag_amd/variants/, "base" = the in-tree build)
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
  echo "== $v"
  python3 tools/probe_gemm_accuracy.py 2>&1 | tail -6
  python3 tools/probe_gemm_step.py 2>&1 | tail -6
  python3 bench.py --no-extras --cpu-seconds 0 --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step', round(d['value']), round(d['ms_per_step'],4), 'loss', d['config'].get('final_loss'))"
done
