#!/bin/bash
# Round 4: the K/V-interleaved arena (SP_KV_INTERLEAVE=1) re-checked on top of the non-temporal gathers (bench.py, graph replay, one box)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4nti}
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-ttft --steps 32 --warmup 8"
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" >> $OUT/ab.txt
  env "${envs[@]}" timeout -k 10 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'attn_ms':r.get('avg_launch_ms'),'attn_GBs':r.get('achieved')}))" >> $OUT/ab.txt || exit 1
}
: > $OUT/ab.txt
for rep in 1 2; do for i in 0 1; do
run headline_il$i SP_KV_INTERLEAVE=$i -- || exit 1
done; done
for i in 0 1; do
run ctx4096_il$i SP_KV_INTERLEAVE=$i -- --ctx 4096 &&
run ctx1024_il$i SP_KV_INTERLEAVE=$i -- --ctx 1024 &&
run bs64_il$i SP_KV_INTERLEAVE=$i -- --bs 64 &&
run r70b_il$i SP_KV_INTERLEAVE=$i -- --model llama3-70b-tp8-rank --bs 128 || exit 1
done
paste - - < $OUT/ab.txt
