#!/bin/bash
# The measurements behind profiles/rNN_extend_w64.txt: the 4-wave x 64-row extend kernel against the 8-wave kernel in one
# process per shape (tools/ab_extend.py), its iteration and workgroup timelines (-DSP_W64_STAMPS / -DSP_W64_WGSTAMPS
# builds: tools/build_w64_variant.sh stamps|wgstamps), and - when a -DSP_W64_PERSIST build is present - the experiment.
# Usage (GPU box): bash tools/report_w64.sh > gpurun_out/extend_w64.txt
set -e
cd "$(dirname "$0")/.."
A=libscratchpad_hip.so@extend_w64=0
cp -f scratchpad_amd/lib/libscratchpad_hip.so scratchpad_amd/lib/libscratchpad_hip_w64.so
B=libscratchpad_hip_w64.so@extend_w64=2
ab() { timeout -k 10 200 python3 tools/ab_extend.py $A $B "$@" 2>&1 | grep -v "amdgpu.ids"; }
echo "# extend attention, bf16, Hq 32 / Hkv 8 / D 128, one MI355X; useful (causal) TFLOP/s, median of 5 rounds x 10 launches"
echo "## config 3 (64 prompts U[128,4096], one launch)"; ab
echo "## 16 x 4096"; ab --bs 16 --len 4096
echo "## 128 x 128 new tokens on an 8192-token prefix (131 interior tiles per row block)"; ab --bs 128 --len 128 --prefix 8192
for spec in "2048 64" "1024 128" "512 256" "256 512" "128 1024" "64 2048"; do set -- $spec; echo "## $1 x $2"; ab --bs $1 --len $2 --rounds 3; done
if [ -f scratchpad_amd/lib/libscratchpad_hip_stamps.so ]; then echo "## one pipelined iteration (s_memtime stamps, long-prefix launch)"; timeout -k 10 100 python3 tools/stamp_w64.py 2>&1 | grep -v amdgpu.ids; fi
if [ -f scratchpad_amd/lib/libscratchpad_hip_wgstamps.so ]; then echo "## a workgroup's life"; timeout -k 10 100 python3 tools/stamp_w64_wg.py 2>&1 | grep -v amdgpu.ids; fi
if [ -f scratchpad_amd/lib/libscratchpad_hip_persist.so ]; then
  echo "## EXPERIMENT: persistent workgroups (-DSP_W64_PERSIST build)"
  P=libscratchpad_hip_persist.so@extend_w64=2,extend_w64_persist=1
  timeout -k 10 200 python3 tools/ab_extend.py $A $B $P 2>&1 | grep -v amdgpu.ids
  timeout -k 10 100 python3 tools/ab_extend.py $A $B $P --bs 2048 --len 64 --rounds 3 2>&1 | grep -v amdgpu.ids
fi
