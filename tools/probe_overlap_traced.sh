#!/bin/bash
# tools/probe_overlap.py under a rocprofv3 kernel trace: were the two branches of its two-stream graphs really in flight
# together?  -> gpurun_out/probe/{probe_traced.txt,probe_trace_overlap.txt} (profiles/r06_overlap_probe.txt, "run 2")
set -o pipefail
R=$GRAFT_REPO_ROOT/gpurun_out/probe
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/trace -o run -- python3 $GRAFT_REPO_ROOT/tools/probe_overlap.py --layers 4 --reps 5 > $R/probe_traced.txt 2>&1 || { tail -20 $R/probe_traced.txt; exit 1; }
cd $GRAFT_REPO_ROOT
TRACE=$(find $R/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py $TRACE > $R/probe_trace_overlap.txt
python tools/trace_overlap.py $TRACE --a decode_mfma_range --b decode_merge >> $R/probe_trace_overlap.txt
rm -rf $R/trace
cat $R/probe_trace_overlap.txt
grep "^stage A\|^micro\|^full" $R/probe_traced.txt
