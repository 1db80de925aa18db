#!/bin/bash
# Round 4: the cap of the decode split size (HipAttnBackend.MAX_CHUNK) 512 vs 1024, in the model (bench.py, graph replay)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4f}
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
run headline_512_$rep SP_DECODE_MAX_CHUNK=512 -- &&
run headline_1024_$rep SP_DECODE_MAX_CHUNK=1024 -- || exit 1
done
run r70b_512 SP_DECODE_MAX_CHUNK=512 -- --model llama3-70b-tp8-rank --bs 128 &&
run r70b_1024 SP_DECODE_MAX_CHUNK=1024 -- --model llama3-70b-tp8-rank --bs 128 &&
run r70b_512_2 SP_DECODE_MAX_CHUNK=512 -- --model llama3-70b-tp8-rank --bs 128 &&
run r70b_1024_2 SP_DECODE_MAX_CHUNK=1024 -- --model llama3-70b-tp8-rank --bs 128 &&
run ctx1024_512 SP_DECODE_MAX_CHUNK=512 -- --ctx 1024 &&
run ctx1024_1024 SP_DECODE_MAX_CHUNK=1024 -- --ctx 1024 &&
run ctx4096_512 SP_DECODE_MAX_CHUNK=512 -- --ctx 4096 &&
run ctx4096_1024 SP_DECODE_MAX_CHUNK=1024 -- --ctx 4096 &&
run bs64_512 SP_DECODE_MAX_CHUNK=512 -- --bs 64 &&
run bs64_1024 SP_DECODE_MAX_CHUNK=1024 -- --bs 64 &&
run bs128_512 SP_DECODE_MAX_CHUNK=512 -- --bs 128 &&
run bs128_1024 SP_DECODE_MAX_CHUNK=1024 -- --bs 128 &&
run fp8_512 SP_DECODE_MAX_CHUNK=512 -- --kv-cache-dtype fp8_e5m2 &&
run fp8_1024 SP_DECODE_MAX_CHUNK=1024 -- --kv-cache-dtype fp8_e5m2
cat $OUT/ab.txt
