#!/bin/bash
# Round 4: where non-temporal K/V gathers start to pay - never (-1) vs always (0) IN THE MODEL (bench.py, graph replay), small launches
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4nts}
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-ttft --steps 48 --warmup 8"
run() { name=$1; shift; mb=$1; shift
  echo "== $name nt_min_mb=$mb" >> $OUT/ab.txt
  SP_BENCH_DEBUG_SET="decode_nt_min_mb=$mb" timeout -k 10 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'attn_ms':r.get('avg_launch_ms'),'attn_GBs':r.get('achieved')}))" >> $OUT/ab.txt || exit 1
}
: > $OUT/ab.txt
for w in "--bs 1 --ctx 4096" "--bs 8 --ctx 1024" "--bs 16 --ctx 1024" "--bs 8 --ctx 4096" "--bs 32 --ctx 1024" "--bs 16 --ctx 4096" "--bs 64 --ctx 1024" "--model llama3-70b-tp8-rank --bs 32" "--model llama3-70b-tp8-rank --bs 64" "--model llama3-70b-tp8-rank --bs 128"; do
  for mb in -1 0 -1 0; do run "$w" $mb $w || exit 1; done
done
paste - - < $OUT/ab.txt
