#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
O=gpurun_out/mkgw; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config mkgw --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
python3 tools/prof_timeline.py $O/trace $O/timeline.txt
rm -rf $O/trace
cat $O/timeline.txt
