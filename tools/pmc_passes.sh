#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only beside it - MI355X_MICROARCH.md, rocprofv3 PMC slots)
# for the extend MFMA kernel and the fp8 decode kernel.  Output: gpurun_out/pmc/summary.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters, program args...
  local name=$1 ctrs=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
EXT="$GRAFT_REPO_ROOT/tools/bench_extend_attn.py --iters 3"
DEC="$GRAFT_REPO_ROOT/tools/bench_decode_attn.py --chunks 512 --iters 4 --warmup 2 --kv fp8"
run ext_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" $EXT &&
run ext_sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $EXT &&
run ext_fetch "FETCH_SIZE" $EXT &&
run dec8_fetch "FETCH_SIZE" $DEC &&
run dec8_write "WRITE_SIZE" $DEC
cd $GRAFT_REPO_ROOT
{
  for d in ext_sq1 ext_sq2 ext_fetch; do echo "## $d"; python tools/pmc_summary.py $OUT/$d extend_mfma_kernel; done
  for d in dec8_fetch dec8_write; do echo "## $d"; python tools/pmc_summary.py $OUT/$d decode_mfma_ decode_merge_kernel; done
  grep -h "extend bs\|chunk" $OUT/ext_fetch.log $OUT/dec8_fetch.log
} > $OUT/summary.txt 2>&1
rm -rf $OUT/ext_sq1 $OUT/ext_sq2 $OUT/ext_fetch $OUT/dec8_fetch $OUT/dec8_write
cat $OUT/summary.txt
