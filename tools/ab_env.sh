#!/bin/bash
# rocprofv3 step timelines under several values of ONE environment variable, one after the other on one box (an engine-level A/B handle such as
# SNAG_TAIL_RIDERS): tools/ab_env.sh VAR value...   prints the step total and the lines matching $AB_GREP
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
VAR=$1; shift
PAT=${AB_GREP:-"gemm_tile_group"}
for v in "$@"; do
  export $VAR=$v
  O=gpurun_out/env_${VAR}_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras ${AB_ARGS:-} > $O/trace.log 2>&1
  python3 tools/prof_timeline.py $O/trace $O/timeline.txt
  rm -rf $O/trace
  echo "== $VAR=$v: $(head -1 $O/timeline.txt)"
  grep "$PAT" $O/timeline.txt | cut -c1-120
done
