#!/bin/bash
# launch timeline (durations, gaps) of the configs[4] scorer entry + two one-thread launches under several libraries: tools/gap_touched.sh NAME...
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
for v in "$@"; do
  if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
  O=gpurun_out/gap_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/probe_touched_gap.py > $O/log.txt 2>&1
  python3 tools/prof_timeline.py $O/trace $O/timeline.txt touched_a_kernel
  rm -rf $O/trace
  echo "== $v"; cat $O/timeline.txt
done
