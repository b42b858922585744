#!/bin/bash
# rocprofv3 kernel table of the drop-in MformerFusion forward + backward (tools/probe_fusion.py): tools/prof_fusion.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
O=gpurun_out/${1:-fusion}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/probe_fusion.py > $O/log.txt 2>&1
python3 tools/prof_summary.py $O/trace $O/kernel_stats.txt > /dev/null
rm -rf $O/trace
grep -v amdgpu.ids $O/log.txt | tail -2; head -24 $O/kernel_stats.txt | cut -c1-90,100-160
