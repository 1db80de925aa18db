#!/bin/bash
# SQ / LDS counters of the decode attention kernel (one counter group per pass, kernel-trace beside it): where the
# waves' cycles go.  -> gpurun_out/pmc_sq/summary.txt
#   bash tools/pmc_decode_sq.sh [NAME "<tools/bench_decode_attn.py shape args>"]...
# default: the bf16 headline shape and config 4's per-rank shape (profiles/r04_decode_pmc_sq.txt); round 5 adds the fp8
# pool: bash tools/pmc_decode_sq.sh head "" fp8 "--kv fp8"
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools/bench_decode_attn.py
run() { local name=$1 ctrs=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/$name -- python3 $T "$@" --iters 4 --warmup 2 --interleave > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G3="GRBM_GUI_ACTIVE GRBM_COUNT"
[ $# -eq 0 ] && set -- head "" hkv1 "--bs 128 --Hq 8 --Hkv 1"
NAMES=()
while [ $# -gt 0 ]; do
  name=$1; args=$2; shift 2
  NAMES+=("$name:$args")
  run ${name}_g1 "$G1" --chunks 768 $args && run ${name}_g2 "$G2" --chunks 768 $args && run ${name}_g3 "$G3" --chunks 768 $args || exit 1
done
cd $GRAFT_REPO_ROOT
{
  echo "# decode attention, tools/bench_decode_attn.py --chunks 768 --interleave (plan + merge launch); default shape = bs 256 Hq 32 Hkv 8 U[128,4096] bf16"
  for na in "${NAMES[@]}"; do
    name=${na%%:*}; echo "# $name: bench_decode_attn.py ${na#*:}"
    for g in g1 g2 g3; do d=${name}_$g; echo "## $d"; python tools/pmc_summary.py $OUT/$d decode_mfma_; grep -h "^chunk" $OUT/$d.log; done
  done
} > $OUT/summary.txt 2>&1
for na in "${NAMES[@]}"; do name=${na%%:*}; for g in g1 g2 g3; do rm -rf $OUT/${name}_$g; done; done
cat $OUT/summary.txt | cut -c1-200
