#!/bin/bash
# Round 4: the host's split policy re-checked with non-temporal gathers in the kernel (bench.py, graph replay, one box)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4ntp}
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
for rep in 1 2; do
for c in 512 768 1024; do
run headline_cap$c SP_DECODE_MAX_CHUNK=$c -- || exit 1
done; done
for c in 512 768 1024; do
run bs64_cap$c SP_DECODE_MAX_CHUNK=$c -- --bs 64 &&
run bs128_cap$c SP_DECODE_MAX_CHUNK=$c -- --bs 128 &&
run r70b_cap$c SP_DECODE_MAX_CHUNK=$c -- --model llama3-70b-tp8-rank --bs 128 &&
run fp8_cap$c SP_DECODE_MAX_CHUNK=$c -- --kv-cache-dtype fp8_e5m2 || exit 1
done
for u in 1 0; do
run ctx1024_uniform$u SP_DECODE_UNIFORM=$u -- --ctx 1024 &&
run ctx4096_uniform$u SP_DECODE_UNIFORM=$u -- --ctx 4096 || exit 1
done
paste - - < $OUT/ab.txt
