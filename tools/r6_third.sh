#!/bin/bash
# round 6: report stage 1 (whole GPU suite, default bench line, prefill line), then the overlap probe under a kernel trace
# (did the two graph branches run at the same time?)
bash tools/gpu_round_report.sh stage1 || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r6
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/probe_trace -o run -- python3 $GRAFT_REPO_ROOT/tools/probe_overlap.py --layers 4 --reps 5 > $R/probe_traced.txt 2>&1 || { tail -20 $R/probe_traced.txt; exit 1; }
cd $GRAFT_REPO_ROOT
TRACE=$(find $R/probe_trace -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py $TRACE > $R/probe_trace_overlap.txt
python tools/trace_overlap.py $TRACE --a decode_mfma_range --b decode_merge >> $R/probe_trace_overlap.txt
rm -rf $R/probe_trace
cat $R/probe_trace_overlap.txt
grep "stage A\|micro\|full" $R/probe_traced.txt
