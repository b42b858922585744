#!/bin/bash
# rocprofv3 step timelines of the MKG-W-shaped step (configs[2]) under several libraries: tools/ab_mkgw.sh NAME...   ("base" = in-tree)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
for v in "$@"; do
  if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
  O=gpurun_out/mkgw_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config mkgw --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
  python3 tools/prof_timeline.py $O/trace $O/timeline.txt
  rm -rf $O/trace
  echo "== $v: $(head -1 $O/timeline.txt)"
  grep "${AB_GREP:-gemm_tile_group_kernel<64\|splitk_reduce_multi}" $O/timeline.txt | cut -c1-120
done
