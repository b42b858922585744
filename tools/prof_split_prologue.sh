#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/splitpro; rm -rf $O; mkdir -p $O
export BENCH_SPLIT_PROLOGUE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
python3 tools/prof_timeline.py $O/trace $O/timeline.txt sample_negatives_kernel
rm -rf $O/trace
head -6 $O/timeline.txt
