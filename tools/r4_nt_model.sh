#!/bin/bash
# Round 4: non-temporal K/V gathers (libscratchpad_hip_nt.so, see tools/r4_nt.sh) against the shipped library IN THE MODEL
# (bench.py, HIP-graph replay), one box.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4ntm}
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-ttft --steps 32 --warmup 8"
run() { name=$1; shift; lib=$1; shift
  echo "== $name ${lib:-shipped}" >> $OUT/ab.txt
  SP_NATIVE_LIB=$lib timeout -k 10 300 $B "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'attn_ms':r.get('avg_launch_ms'),'attn_GBs':r.get('achieved')}))" >> $OUT/ab.txt || exit 1
}
: > $OUT/ab.txt
NT=libscratchpad_hip_nt.so
for lib in "" $NT "" $NT; do run headline "$lib" || exit 1; done
for w in "--ctx 128" "--ctx 1024" "--ctx 4096" "--bs 8 --ctx 1024" "--bs 32" "--bs 64" "--bs 128" "--model llama3-70b-tp8-rank --bs 128" "--kv-cache-dtype fp8_e5m2"; do
  for lib in "" $NT; do run "$w" "$lib" $w || exit 1; done
done
paste - - < $OUT/ab.txt
