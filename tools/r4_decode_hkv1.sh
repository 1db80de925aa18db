#!/bin/bash
# Round 4: decode attention at config 4's per-rank shape (bs 128, Hq 8, Hkv 1, D 128, bf16) next to the headline shape:
# separate K / V arenas vs one interleaved arena, then rocprofv3 kernel trace and FETCH/WRITE PMC passes of the Hkv=1 shape.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4a}
mkdir -p $OUT
T=$GRAFT_REPO_ROOT/tools/bench_decode_attn.py
H1="--bs 128 --Hq 8 --Hkv 1"
{
echo "== headline separate";   python3 $T --chunks 512 --iters 50 --warmup 20 --check
echo "== headline interleave"; python3 $T --chunks 512 --iters 50 --warmup 20 --check --interleave
echo "== hkv1 separate";       python3 $T $H1 --chunks 64,128,256,512 --iters 50 --warmup 20 --check
echo "== hkv1 interleave";     python3 $T $H1 --chunks 64,128,256,512 --iters 50 --warmup 20 --check --interleave
echo "== hkv2 (tp4) separate";   python3 $T --bs 128 --Hq 16 --Hkv 2 --chunks 256,512 --iters 50 --warmup 20
echo "== hkv2 (tp4) interleave"; python3 $T --bs 128 --Hq 16 --Hkv 2 --chunks 256,512 --iters 50 --warmup 20 --interleave
echo "== ctx128 separate";   python3 $T --ctx 128 --chunks 128 --iters 50 --warmup 20
echo "== ctx128 interleave"; python3 $T --ctx 128 --chunks 128 --iters 50 --warmup 20 --interleave
} > $OUT/ab.txt 2>&1 || { tail -20 $OUT/ab.txt; exit 1; }
cat $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
DEC="$T $H1 --chunks 512 --iters 8 --warmup 4"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $DEC > $OUT/kt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $DEC > $OUT/f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $DEC > $OUT/w.log 2>&1
cd $GRAFT_REPO_ROOT
{
find $OUT/kt -name "*kernel_stats.csv" -exec cat {} \; | grep -i "decode\|Name" 
python tools/pmc_summary.py $OUT/f decode_
python tools/pmc_summary.py $OUT/w decode_
grep -h chunk $OUT/f.log
} > $OUT/summary.txt 2>&1
rm -rf $OUT/f $OUT/w
find $OUT/kt -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt
