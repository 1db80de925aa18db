#!/bin/bash
# round 6: plan-kernel timing, report stage 3 (mllama, gemv, 70B TP 4 rehearsal at full depth) and the rocprofv3 kernel traces
mkdir -p gpurun_out/r6
timeout -k 10 200 python tools/bench_decode_plan.py > gpurun_out/r6/decode_plan_us.txt 2>&1 || { tail -5 gpurun_out/r6/decode_plan_us.txt; exit 1; }
grep sp_decode_plan gpurun_out/r6/decode_plan_us.txt
bash tools/gpu_round_report.sh --profile-only || exit 1
bash tools/gpu_round_report.sh stage3 || exit 1
