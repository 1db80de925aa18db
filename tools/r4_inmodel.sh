#!/bin/bash
# Round 4: in-model A/B of the interleaved K/V arena and of the fused split merge (bench.py, HIP-graph replay)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c}
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-ttft --steps 32 --warmup 8"
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" >> $OUT/ab.txt
  env "${envs[@]}" timeout -k 10 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'attn_ms':r.get('avg_launch_ms'),'attn_GBs':r.get('achieved'),'frac':r.get('frac')}))" >> $OUT/ab.txt || exit 1
}
: > $OUT/ab.txt
run headline_interleave SP_KV_INTERLEAVE=1 -- &&
run headline_separate SP_KV_INTERLEAVE=0 -- &&
run headline_interleave_2 SP_KV_INTERLEAVE=1 -- &&
run headline_separate_2 SP_KV_INTERLEAVE=0 -- &&
run ctx128_interleave SP_KV_INTERLEAVE=1 -- --ctx 128 &&
run ctx128_separate SP_KV_INTERLEAVE=0 -- --ctx 128 &&
run r70b_interleave SP_KV_INTERLEAVE=1 -- --model llama3-70b-tp8-rank --bs 128 &&
run r70b_separate SP_KV_INTERLEAVE=0 -- --model llama3-70b-tp8-rank --bs 128 &&
run r70b_fused SP_DECODE_FUSE_MERGE=1 -- --model llama3-70b-tp8-rank --bs 128 &&
run headline_fused SP_DECODE_FUSE_MERGE=1 -- &&
run bs1_fused SP_DECODE_FUSE_MERGE=1 -- --bs 1 --ctx 1024 &&
run bs1_unfused SP_DECODE_FUSE_MERGE=0 -- --bs 1 --ctx 1024 &&
run bs8_fused SP_DECODE_FUSE_MERGE=1 -- --bs 8 --ctx 1024 &&
run bs8_unfused SP_DECODE_FUSE_MERGE=0 -- --bs 8 --ctx 1024 &&
run bs32_fused SP_DECODE_FUSE_MERGE=1 -- --bs 32 --ctx 1024 &&
run bs32_unfused SP_DECODE_FUSE_MERGE=0 -- --bs 32 --ctx 1024
cat $OUT/ab.txt
timeout -k 10 700 python -m pytest tests/test_gpu_tensor_parallel.py tests/test_gpu_schedule_flow.py tests/test_gpu_llama.py tests/test_gpu_long_context.py tests/test_gpu_mllama.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
