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
for c in 256 128 512; do
run bs32ctx1024_t$c SP_DECODE_TARGET_ITEMS=$c -- --bs 32 --ctx 1024 &&
run bs16ctx4096_t$c SP_DECODE_TARGET_ITEMS=$c -- --bs 16 --ctx 4096 &&
run bs8ctx4096_t$c SP_DECODE_TARGET_ITEMS=$c -- --bs 8 --ctx 4096 &&
run bs64ctx1024_t$c SP_DECODE_TARGET_ITEMS=$c -- --bs 64 --ctx 1024 &&
run bs32_t$c SP_DECODE_TARGET_ITEMS=$c -- --bs 32 || exit 1
done
paste - - < $OUT/ab.txt
