#!/bin/bash
# Round 4: non-temporal K/V gathers in the decode kernel (diagnostic library libscratchpad_hip_nt.so, built from a copy of
# decode_mfma.hip whose two gather loads are __builtin_nontemporal_load) against the shipped library, kernel alone, one box.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4nt}
mkdir -p $OUT
: > $OUT/nt.txt
T="python3 tools/bench_decode_attn.py --no-fuse --iters 30"
for rep in 1 2; do
for lib in "" libscratchpad_hip_nt.so; do
  L=${lib:+--lib $lib}
  echo "#### rep $rep lib ${lib:-shipped}" >> $OUT/nt.txt
  echo "== headline, chunk 768" >> $OUT/nt.txt; timeout -k 10 200 $T --chunks 768 $L 2>&1 | grep -E "chunk|median" >> $OUT/nt.txt || exit 1
  echo "== headline, chunk 768, GEMM between launches" >> $OUT/nt.txt; timeout -k 10 200 $T --chunks 768 --gemm $L 2>&1 | grep -E "chunk|median" >> $OUT/nt.txt || exit 1
  echo "== Hq 8 / Hkv 1 bs 128, chunk 768" >> $OUT/nt.txt; timeout -k 10 200 $T --bs 128 --Hq 8 --Hkv 1 --chunks 768 $L 2>&1 | grep -E "chunk|median" >> $OUT/nt.txt || exit 1
  echo "== ctx 4096 unsplit" >> $OUT/nt.txt; timeout -k 10 200 $T --ctx 4096 --chunks 4160 $L 2>&1 | grep -E "chunk|median" >> $OUT/nt.txt || exit 1
  echo "== bs 32 ctx 1024 chunk 256" >> $OUT/nt.txt; timeout -k 10 200 $T --bs 32 --ctx 1024 --chunks 256 $L 2>&1 | grep -E "chunk|median" >> $OUT/nt.txt || exit 1
done; done
cat $OUT/nt.txt
