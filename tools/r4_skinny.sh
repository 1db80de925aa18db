#!/bin/bash
# Round 4: SiluAndMul fused into the <= 16-row down projection - parity, kernel times, and the small-batch steps
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4e}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_llama.py tests/test_gpu_schedule_flow.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python tools/bench_gemv.py 2>&1 | grep -v amdgpu.ids > $OUT/gemv.log || { tail $OUT/gemv.log; exit 1; }
cat $OUT/gemv.log
B="python3 bench.py --no-cpu-baseline --no-ttft --steps 64 --warmup 8 --ctx 1024"
for bs in 1 8 16; do for f in 1 0; do
echo "== bs $bs SP_SKINNY_SILU=$f"; SP_SKINNY_SILU=$f timeout -k 10 200 $B --bs $bs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms/step')" || exit 1
done; done 2>&1 | tee $OUT/steps.txt
