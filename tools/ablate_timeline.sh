#!/bin/bash
# Step timelines of the -DSNAG_GEMM_ABLATE=1 variant under several SNAG_GEMM_DBG masks (results are wrong; timing only):
#   tools/build_variant.sh ablate -DSNAG_GEMM_ABLATE=1 ; tools/gpu.sh 'bash tools/ablate_timeline.sh 0 8 16 24 3 56'
# bits: 1 no A traffic, 2 no B traffic (k-tiles >= 2 read zeros), 4 no barrier, 8 no MFMAs, 16 no split / image write, 32 no fetch, 64 no B split (ping-pong loop)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_ablate.so
PAT=${AB_GREP:-"gemm_tile"}
for m in "$@"; do
  export SNAG_GEMM_DBG=$m
  O=gpurun_out/ablate_$m; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
  python3 tools/prof_timeline.py $O/trace $O/timeline.txt
  rm -rf $O/trace
  echo "== DBG=$m: $(head -1 $O/timeline.txt)"
  grep "$PAT" $O/timeline.txt | cut -c1-130
done
