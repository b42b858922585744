#!/bin/bash
# rocprofv3 kernel stats of bench.py --config stress (uniform and Zipf): tools/prof_stress.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
TAG=${1:-r03_stress}
for dist in uniform zipf; do
  O=gpurun_out/${TAG}_$dist
  rm -rf $O && mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config stress --index-dist $dist --steps 30 --warmup 5 --cpu-seconds 0 > $O/trace.log 2>&1
  python3 tools/prof_summary.py $O/trace $O/kernel_stats.txt > /dev/null
  grep metric $O/trace.log > $O/bench_under_rocprof.json
  rm -rf $O/trace
  python3 bench.py --config stress --index-dist $dist --cpu-seconds 0 2>/dev/null | grep metric > $O/bench.json
  head -20 $O/kernel_stats.txt
done
