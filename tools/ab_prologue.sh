#!/bin/bash
# A/B on the GPU box: step prologue fused vs. the sampler as its own launch
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  python bench.py --no-extras --cpu-seconds 0 2>/dev/null | p fused
  BENCH_SPLIT_PROLOGUE=1 python bench.py --no-extras --cpu-seconds 0 2>/dev/null | p split
done
