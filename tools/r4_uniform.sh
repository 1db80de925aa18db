#!/bin/bash
# Round 4: the near-uniform split policy (advisory seq_lens_max_hint) in the model: hint on (default) vs off
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4i}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_schedule_flow.py tests/test_gpu_llama.py tests/test_gpu_long_context.py tests/test_gpu_plan_overflow.py tests/test_gpu_mllama.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
B="python3 bench.py --no-cpu-baseline --no-ttft --steps 32 --warmup 8"
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" >> $OUT/ab.txt
  env "${envs[@]}" timeout -k 10 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'attn_ms':r.get('avg_launch_ms'),'attn_GBs':r.get('achieved')}))" >> $OUT/ab.txt || exit 1
}
: > $OUT/ab.txt
for h in 1 0; do
run ctx1024_hint$h SP_DECODE_UNIFORM=$h -- --ctx 1024 &&
run ctx4096_hint$h SP_DECODE_UNIFORM=$h -- --ctx 4096 &&
run ctx128_hint$h SP_DECODE_UNIFORM=$h -- --ctx 128 &&
run bs128ctx1024_hint$h SP_DECODE_UNIFORM=$h -- --bs 128 --ctx 1024 &&
run bs64ctx4096_hint$h SP_DECODE_UNIFORM=$h -- --bs 64 --ctx 4096 &&
run bs32ctx1024_hint$h SP_DECODE_UNIFORM=$h -- --bs 32 --ctx 1024 &&
run headline_hint$h SP_DECODE_UNIFORM=$h -- &&
run r70bctx4096_hint$h SP_DECODE_UNIFORM=$h -- --model llama3-70b-tp8-rank --bs 128 --ctx 4096 || exit 1
done
paste - - < $OUT/ab.txt
