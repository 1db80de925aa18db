#!/bin/bash
# SQ / LDS counters of the bf16 decode attention kernel at the headline shape and at config 4's per-rank shape
# (one counter group per pass, kernel-trace beside it): where the waves' cycles go.  -> gpurun_out/pmc_sq/summary.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools/bench_decode_attn.py
run() { local name=$1 ctrs=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/$name -- python3 $T "$@" --iters 4 --warmup 2 --no-fuse > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G3="GRBM_GUI_ACTIVE GRBM_COUNT"
run head_g1 "$G1" --chunks 768 && run head_g2 "$G2" --chunks 768 && run head_g3 "$G3" --chunks 768 &&
run hkv1_g1 "$G1" --bs 128 --Hq 8 --Hkv 1 --chunks 768 && run hkv1_g2 "$G2" --bs 128 --Hq 8 --Hkv 1 --chunks 768 && run hkv1_g3 "$G3" --bs 128 --Hq 8 --Hkv 1 --chunks 768 || exit 1
cd $GRAFT_REPO_ROOT
{
  echo "# bf16 decode attention, tools/bench_decode_attn.py --chunks 768 --no-fuse; headline = bs 256 Hq 32 Hkv 8 U[128,4096]; hkv1 = bs 128 Hq 8 Hkv 1"
  for d in head_g1 head_g2 head_g3 hkv1_g1 hkv1_g2 hkv1_g3; do echo "## $d"; python tools/pmc_summary.py $OUT/$d decode_mfma_kernel; grep -h "^chunk" $OUT/$d.log; done
} > $OUT/summary.txt 2>&1
for d in head_g1 head_g2 head_g3 hkv1_g1 hkv1_g2 hkv1_g3; do rm -rf $OUT/$d; done
cat $OUT/summary.txt | cut -c1-200
