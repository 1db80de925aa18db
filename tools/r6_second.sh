#!/bin/bash
# round 6: the whole GPU suite on the ABI-9 tree, then the default bench line
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > gpurun_out/r6/tests_full.log 2>&1
rc=$?
grep -h "passed\|failed" gpurun_out/r6/tests_full.log | tail -2
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r6/tests_full.log; fi
if [ $rc -le 1 ]; then
  timeout -k 10 250 python bench.py > gpurun_out/r6/bench_decode.json 2> gpurun_out/r6/bench_decode.err
  rc2=$?
  tail -1 gpurun_out/r6/bench_decode.json | cut -c1-1200
  [ $rc2 -ne 0 ] && tail -20 gpurun_out/r6/bench_decode.err
  exit $(( rc > rc2 ? rc : rc2 ))
fi
exit $rc
