#!/bin/bash
# PMC passes for the extend (prefill) MFMA kernel, one counter group per run (MI355X_MICROARCH.md,
# rocprofv3 PMC slots).  Output: gpurun_out/pmc_ext/summary.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ext
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters, program args...
  local name=$1 ctrs=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
EXT="$GRAFT_REPO_ROOT/tools/bench_extend_attn.py --iters 3 --rounds 1"
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" $EXT &&
run sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" $EXT &&
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $EXT &&
run fetch "FETCH_SIZE" $EXT &&
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT" $EXT
cd $GRAFT_REPO_ROOT
{
  for d in sq1 sq2 tcc fetch grbm; do echo "## $d"; python tools/pmc_summary.py $OUT/$d extend_mfma_kernel; done
  grep -h "extend bs" $OUT/fetch.log
} > $OUT/summary.txt 2>&1
python tools/pmc_clock.py $OUT/grbm extend_mfma_kernel >> $OUT/summary.txt 2>&1
rm -rf $OUT/sq1 $OUT/sq2 $OUT/tcc $OUT/fetch $OUT/grbm
cat $OUT/summary.txt
