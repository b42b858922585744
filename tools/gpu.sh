#!/bin/bash
# build the library here (hipcc cross-compiles), then run the command on a GPU box:  tools/gpu.sh [timeout_s] 'command'
set -e
cd "$(dirname "$0")/.."
T=1500
if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
python -m snag_amd.build | tail -1
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
