#!/bin/bash
# A/B build that recompiles ONE source with extra -D switches and links it with the in-tree objects of the others:
#   tools/build_variant_one.sh NAME readout_fused -DSNAG_RO_ABLATE=1   -> snag_amd/variants/libsnag_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
python -m snag_amd.build > /dev/null
mkdir -p snag_amd/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off -fvisibility=hidden "$@" -c snag_amd/csrc/$SRC.hip -o snag_amd/variants/${SRC}_$NAME.o
OBJS=$(ls snag_amd/csrc/_obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o snag_amd/variants/libsnag_hip_$NAME.so $OBJS snag_amd/variants/${SRC}_$NAME.o
rm snag_amd/variants/${SRC}_$NAME.o
echo built snag_amd/variants/libsnag_hip_$NAME.so
