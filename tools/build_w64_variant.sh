#!/bin/bash
# A/B builds of the 4 x 64-row extend kernel: tools/build_w64_variant.sh NAME "-DFLAG ..." compiles extend_w64.hip alone
# with the given defines and links it with the shipped objects into scratchpad_amd/lib/libscratchpad_hip_NAME.so
# (tools/ab_extend.py LIB@extend_w64=1 ... times several libraries / switches in one process).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fvisibility=hidden -Wno-unused-value -fno-honor-nans \
  -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -c scratchpad_amd/csrc/extend_w64.hip -o build/variants/extend_w64_$NAME.o
OBJS=$(ls build/obj/*.o | grep -v extend_w64.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratchpad_amd/lib/libscratchpad_hip_$NAME.so $OBJS build/variants/extend_w64_$NAME.o
python3 tools/patch_w64_descriptor.py scratchpad_amd/lib/libscratchpad_hip_$NAME.so
echo built scratchpad_amd/lib/libscratchpad_hip_$NAME.so
