#!/bin/bash
# A/B builds of the extend kernel: tools/build_variant.sh NAME "-DFLAG ..." compiles extend_mfma.hip alone with the
# given defines (headline instantiation only: seconds) and links it with the shipped objects into
# scratchpad_amd/lib/libscratchpad_hip_NAME.so; tools/ab_extend.py times several such libraries in one process.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -fno-honor-nans \
  -DSP_EXTEND_ONLY_HEADLINE "$@" -c scratchpad_amd/csrc/extend_mfma.hip -o build/variants/extend_mfma_$NAME.o
OBJS=$(ls build/obj/*.o | grep -v extend_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratchpad_amd/lib/libscratchpad_hip_$NAME.so $OBJS build/variants/extend_mfma_$NAME.o
python3 tools/w64_asm.py check scratchpad_amd/lib/libscratchpad_hip_$NAME.so > /dev/null   # (build/obj/extend_w64.o comes from the staged build: build.py compile_w64)
echo built scratchpad_amd/lib/libscratchpad_hip_$NAME.so
