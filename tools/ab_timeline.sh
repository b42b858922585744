#!/bin/bash
# rocprofv3 step timelines of several variant libraries, one after the other on one box: tools/ab_timeline.sh NAME... ("base" = the in-tree library)
# prints the step total and the lines of the kernels matching $AB_GREP (default: the 256-row grouped GEMM launches)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
PAT=${AB_GREP:-"gemm_tile_group_kernel<256"}
for v in "$@"; do
  if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
  O=gpurun_out/var_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
  python3 tools/prof_timeline.py $O/trace $O/timeline.txt
  rm -rf $O/trace
  echo "== $v: $(head -1 $O/timeline.txt)"
  grep "$PAT" $O/timeline.txt | cut -c1-120
done
