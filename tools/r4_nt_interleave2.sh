#!/bin/bash
# Round 4: interleaved arena on top of the non-temporal gathers, second box: headline x3, ctx 1024 x2, fp8 pool, bs 32 / 128, the TTFT pass
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4nti2}
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
for rep in 1 2 3; do for i in 0 1; do run headline_il$i SP_KV_INTERLEAVE=$i -- || exit 1; done; done
for rep in 1 2; do for i in 0 1; do run ctx1024_il$i SP_KV_INTERLEAVE=$i -- --ctx 1024 || exit 1; done; done
for i in 0 1; do
run fp8_il$i SP_KV_INTERLEAVE=$i -- --kv-cache-dtype fp8_e5m2 &&
run bs32_il$i SP_KV_INTERLEAVE=$i -- --bs 32 &&
run bs128_il$i SP_KV_INTERLEAVE=$i -- --bs 128 &&
run bs8_il$i SP_KV_INTERLEAVE=$i -- --bs 8 --ctx 1024 || exit 1
done
paste - - < $OUT/ab.txt
for i in 0 1 0 1; do echo "== prefill il$i"; SP_KV_INTERLEAVE=$i timeout -k 10 300 python3 bench.py --mode prefill --steps 2 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline_prefill']
print('  ttft_p50', d['value'], 'pass', d['ms_per_step'], 'attn TFLOP/s', r['achieved'], r['avg_launch_ms'])" || exit 1; done
