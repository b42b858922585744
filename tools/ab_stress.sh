#!/bin/bash
# configs[4] scorer kernels under several libraries: tools/ab_stress.sh NAME...   ("base" = in-tree); prints the touched_* kernel averages, uniform and Zipf
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
for v in "$@"; do
  if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
  for dist in uniform zipf; do
    O=gpurun_out/abs_${v}_$dist; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config stress --index-dist $dist --steps 30 --warmup 5 --cpu-seconds 0 > $O/trace.log 2>&1
    python3 tools/prof_summary.py $O/trace $O/kernel_stats.txt > /dev/null
    rm -rf $O/trace
    echo "== $v $dist: $(grep -o '"score_fwd_bwd[^,]*' $O/trace.log | head -1)"
    grep "touched_" $O/kernel_stats.txt | awk '{print "   ", $0}' | cut -c1-60,105-150
  done
done
