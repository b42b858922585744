#!/bin/bash
# Build an A/B variant of the library with extra -D switches: tools/build_variant.sh NAME -DSNAG_GEMM_PIPE=1 ...
# -> snag_amd/variants/libsnag_hip_NAME.so (git-ignored; select it with SNAG_HIP_LIB=<path>).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p snag_amd/variants/obj_$NAME
for f in snag_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off -fvisibility=hidden "$@" -c $f -o snag_amd/variants/obj_$NAME/$b.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o snag_amd/variants/libsnag_hip_$NAME.so snag_amd/variants/obj_$NAME/*.o
rm -rf snag_amd/variants/obj_$NAME
echo built snag_amd/variants/libsnag_hip_$NAME.so
