#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
O=gpurun_out/mmea; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/probe_mmea.py > $O/log 2>&1
python3 tools/prof_by_shape.py $O/t $O/by_shape.txt > /dev/null
head -45 $O/by_shape.txt
rm -rf $O/t
