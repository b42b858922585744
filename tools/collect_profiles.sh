#!/bin/bash
# Run on the GPU box (gpurun): kernel trace + the two HBM PMC passes (separate runs, as the guide prescribes).
#   tools/collect_profiles.sh <tag> [pmc]     (without "pmc": kernel trace only)
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
TAG=${1:-r02}
O=gpurun_out/$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
python3 tools/prof_summary.py $O/trace $O/kernel_stats.txt > /dev/null
python3 tools/prof_by_shape.py $O/trace $O/by_shape.txt > /dev/null
python3 tools/prof_timeline.py $O/trace $O/timeline.txt
grep metric $O/trace.log > $O/bench_under_rocprof.json
if [ "${2:-}" = "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/pmc_write.log 2>&1
  python3 tools/prof_by_shape.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm.txt > /dev/null
  rm -rf $O/pmc_fetch $O/pmc_write
fi
python3 bench.py > $O/bench.log 2>&1; grep metric $O/bench.log > $O/bench.json
rm -rf $O/trace
ls -la $O
