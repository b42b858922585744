#!/bin/bash
# cycle stamps of the one-launch read-out forward (variant built with -DSNAG_RO_STAMP=1): tools/build_variant_one.sh stamp readout_fused -DSNAG_RO_STAMP=1
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
export SNAG_HIP_LIB=$PWD/snag_amd/variants/libsnag_hip_stamp.so
python3 bench.py --steps 6 --warmup 3 --cpu-seconds 0 --no-kernel-timers --no-extras 2>&1 | grep -A9 "ro stamps" | tail -10
