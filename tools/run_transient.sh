#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
rm -rf gpurun_out/trans && mkdir -p gpurun_out/trans
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trans/t -- python3 bench.py --steps 80 --warmup 0 --cpu-seconds 0 --no-extras --no-kernel-timers > gpurun_out/trans/log 2>&1
python3 tools/prof_transient.py gpurun_out/trans/t | tee gpurun_out/trans/steps.txt | head -60
rm -rf gpurun_out/trans/t
