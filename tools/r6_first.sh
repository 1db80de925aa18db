#!/bin/bash
# round 6, first GPU call: the attention / plan tests touched by ABI 9, then the overlap probe
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_plan_overflow.py tests/test_gpu_schedule_flow.py -x -q -m gpu -s > gpurun_out/r6/t1.log 2>&1
rc=$?
tail -5 gpurun_out/r6/t1.log
if [ $rc -le 1 ]; then
  timeout -k 10 600 python tools/probe_overlap.py --out gpurun_out/r6/probe_overlap.json > gpurun_out/r6/probe_overlap.txt 2>&1
  rc2=$?
  cat gpurun_out/r6/probe_overlap.txt | tail -20
  exit $(( rc > rc2 ? rc : rc2 ))
fi
exit $rc
