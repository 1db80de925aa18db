#!/bin/bash
# Diagnostic twin of the library with in-kernel cycle stamps in the extend kernel
# (build/libscratchpad_hip_stamps.so; loaded only by tools/stamp_extend_attn.py).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamps
for f in scratchpad_amd/csrc/*.hip; do
  [ "$(basename $f)" = "extend_w64.hip" ] && continue      # needs the staged compile: the shipped object is linked below
  extra=""
  case $(basename $f) in
    elementwise.hip) extra="-ffp-contract=off";;
    extend_mfma.hip) extra="-fno-honor-nans";;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value \
    ${SP_STAMP_DEFS:--DSP_EXTEND_STAMPS} $extra -c $f -o build/stamps/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratchpad_amd/lib/libscratchpad_hip_stamps.so build/stamps/*.o build/obj/extend_w64.o
python3 tools/w64_asm.py check scratchpad_amd/lib/libscratchpad_hip_stamps.so > /dev/null
echo built scratchpad_amd/lib/libscratchpad_hip_stamps.so
