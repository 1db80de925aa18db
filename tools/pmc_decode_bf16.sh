set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DEC="$GRAFT_REPO_ROOT/tools/bench_decode_attn.py --chunks 512 --iters 4 --warmup 2"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $DEC > $OUT/f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $DEC > $OUT/w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/f decode_mfma_kernel decode_merge_kernel > $OUT/summary.txt
python tools/pmc_summary.py $OUT/w decode_mfma_kernel decode_merge_kernel >> $OUT/summary.txt
grep -h chunk $OUT/f.log >> $OUT/summary.txt
rm -rf $OUT/f $OUT/w
cat $OUT/summary.txt
