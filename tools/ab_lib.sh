#!/bin/bash
# A/B two library builds on one device: tools/ab_lib.sh <variant .so> -- runs the GEMM shape probes and the step bench with each.
cd "$GRAFT_REPO_ROOT"
V=$1
for lib in "" "$PWD/$V" "" "$PWD/$V"; do
  if [ -n "$lib" ]; then export SNAG_HIP_LIB=$lib; tag=variant; else unset SNAG_HIP_LIB; tag=base; fi
  echo "== $tag"
  TILES=0 python3 tools/tune_gemm_mmea.py | cut -c1-700
  SNAG_GEMM_XCD=1 python3 tools/ab_xcd.py child
  python3 bench.py --no-extras --cpu-seconds 0 --steps 100 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['avg_us'],1), round(d['roofline']['frac'],3))"
done
