#!/bin/bash
# rocprofv3 kernel table of the configs[3] evaluation (tools/probe_eval.py): tools/prof_eval.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
O=gpurun_out/${1:-eval}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/probe_eval.py > $O/log.txt 2>&1
python3 tools/prof_summary.py $O/trace $O/kernel_stats.txt > /dev/null
rm -rf $O/trace
grep -v amdgpu.ids $O/log.txt; head -14 $O/kernel_stats.txt | cut -c1-75,100-160
