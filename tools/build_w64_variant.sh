#!/bin/bash
# A/B builds of the 4 x 64-row extend kernel: tools/build_w64_variant.sh NAME "-DFLAG ..." compiles extend_w64.hip alone
# with the given defines and links it with the shipped objects into scratchpad_amd/lib/libscratchpad_hip_NAME.so
# (tools/ab_extend.py LIB@extend_w64=1 ... times several libraries / switches in one process).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/variants
# the staged compile of this file (device assembly -> tools/w64_asm.py -> assembler -> host object): build.py compile_w64
python3 -m scratchpad_amd.build --w64-object build/variants/extend_w64_$NAME.o "$@" > /dev/null
OBJS=$(ls build/obj/*.o | grep -v extend_w64.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratchpad_amd/lib/libscratchpad_hip_$NAME.so $OBJS build/variants/extend_w64_$NAME.o
python3 tools/w64_asm.py check scratchpad_amd/lib/libscratchpad_hip_$NAME.so
echo built scratchpad_amd/lib/libscratchpad_hip_$NAME.so
