#!/bin/bash
# rocprofv3 timeline of the step with a variant library: tools/prof_variant.sh NAME
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$1.so
O=gpurun_out/var_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
python3 tools/prof_timeline.py $O/trace $O/timeline.txt
rm -rf $O/trace
cat $O/timeline.txt | cut -c1-100
