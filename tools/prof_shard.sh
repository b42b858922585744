#!/bin/bash
# one rank's share of an 8-way row-sharded step (projections + Mformer over 1875 entities, scoring at the full batch), no collectives
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
O=gpurun_out/shard8; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/probe_host_rate.py --rows-div ${1:-8} --steps 40 > $O/trace.log 2>&1
python3 tools/prof_timeline.py $O/trace $O/timeline.txt
rm -rf $O/trace
cat $O/timeline.txt
