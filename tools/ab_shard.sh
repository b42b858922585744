#!/bin/bash
# one rank's share of an R-way row-sharded step under several libraries: tools/ab_shard.sh "R..." NAME...   ("base" = in-tree)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
RS=$1; shift
for R in $RS; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SNAG_HIP_LIB; else export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_$v.so; fi
    O=gpurun_out/shard_${R}_$v; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/probe_host_rate.py --rows-div $R --steps 40 > $O/trace.log 2>&1
    python3 tools/prof_timeline.py $O/trace $O/timeline.txt
    rm -rf $O/trace
    echo "== R=$R $v: $(head -1 $O/timeline.txt)"
    grep "${AB_GREP:-readout\|attn_ro\|gemm_tile_ln\|gemm_tile_kernel\|gemm_tile_group_kernel<64, 128, 32, 64, true, true}" $O/timeline.txt | cut -c1-120
  done
done
