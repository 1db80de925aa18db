#!/bin/bash
# The measurements behind profiles/rNN_extend_w64.txt: the 4-wave x 64-row extend kernel - one workgroup per item, and its
# persistent form - against the 8-wave kernel in one process per shape (tools/ab_extend.py), the iteration and workgroup
# timelines (-DSP_W64_STAMPS / -DSP_W64_WGSTAMPS builds: tools/build_w64_variant.sh w64stamps|wgstamps).
# Usage (GPU box): bash tools/report_w64.sh > gpurun_out/extend_w64.txt
set -e
cd "$(dirname "$0")/.."
# (two specs of one FILE would share its switches: copies)
cp -f scratchpad_amd/lib/libscratchpad_hip.so scratchpad_amd/lib/libscratchpad_hip_w64.so
cp -f scratchpad_amd/lib/libscratchpad_hip.so scratchpad_amd/lib/libscratchpad_hip_w64p.so
cp -f scratchpad_amd/lib/libscratchpad_hip.so scratchpad_amd/lib/libscratchpad_hip_auto.so
A=libscratchpad_hip.so@extend_w64=0
B=libscratchpad_hip_w64.so@extend_w64=2,extend_w64_persist=0
P=libscratchpad_hip_w64p.so@extend_w64=2,extend_w64_persist=2
S=libscratchpad_hip_auto.so@extend_w64=1,extend_w64_persist=1
ab() { timeout -k 10 300 python3 tools/ab_extend.py $A $B $P $S "$@" 2>&1 | grep -v "amdgpu.ids\|max |diff| = 0.000e+00"; }
echo "# extend attention, bf16, Hq 32 / Hkv 8 / D 128, one MI355X; useful (causal) TFLOP/s, median of 5 rounds x 10 launches, all four in one process"
echo "# extend_w64=0: 8-wave kernel; =2,persist=0: w64, one workgroup per item; =2,persist=2: w64, persistent workgroups; =1,persist=1: what ships (picked by launch shape)"
echo "# (outputs compared with the first library's: lines of identical outputs removed)"
echo "## config 3 (64 prompts U[128,4096], one launch)"; ab
echo "## 16 prompts U[128,4096]"; ab --bs 16
for spec in "4096 32" "2048 64" "1024 128" "512 256" "256 512" "128 1024" "64 2048" "16 4096" "4 8192" "1 4096" "8 300"; do set -- $spec; echo "## $1 x $2"; ab --bs $1 --len $2 --rounds 3; done
echo "## 128 x 128 new tokens on an 8192-token prefix (131 interior tiles per row block)"; ab --bs 128 --len 128 --prefix 8192
echo "## 256 x 64 new tokens on a 2048-token prefix"; ab --bs 256 --len 64 --prefix 2048 --rounds 3
echo "## 32 x 512 new tokens on a 512-token prefix"; ab --bs 32 --len 512 --prefix 512 --rounds 3
if [ -f scratchpad_amd/lib/libscratchpad_hip_w64stamps.so ]; then echo "## one pipelined iteration (s_memtime stamps, long-prefix launch, one workgroup per item)"; timeout -k 10 100 python3 tools/stamp_w64.py 2>&1 | grep -v amdgpu.ids; fi
if [ -f scratchpad_amd/lib/libscratchpad_hip_wgstamps.so ]; then
  echo "## a workgroup's life (one workgroup per item)"; timeout -k 10 100 python3 tools/stamp_w64_wg.py libscratchpad_hip_wgstamps.so@extend_w64=2,extend_w64_persist=0 2>&1 | grep -v amdgpu.ids
  echo "## an item's life in a persistent workgroup"; timeout -k 10 100 python3 tools/stamp_w64_wg.py libscratchpad_hip_wgstamps.so@extend_w64=2,extend_w64_persist=2 persist 2>&1 | grep -v amdgpu.ids
fi
