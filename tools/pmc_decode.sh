#!/bin/bash
# PMC traffic of the decode attention kernels for ONE workload: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots) plus a --kernel-trace --stats pass, then the traffic / algorithmic
# ratio and the hash of the kernel sources it was measured on.
#   bash tools/pmc_decode.sh NAME BENCH_WORKLOAD_KEY "<tools/bench_decode_attn.py shape arguments>"
#   e.g.  bash tools/pmc_decode.sh hkv1 "llama3-70b-tp8-rank|bs128|ctxuniform|kvauto" "--bs 128 --Hq 8 --Hkv 1 --chunks 768 --interleave"
# -> gpurun_out/pmc_NAME/{summary.txt,decode_attn_pmc_NAME.json,kernel_stats.csv}: copy to profiles/rNN_decode_attn_pmc_NAME.*
# (bench.py's pmc_traffic() uses the ratio only for the same BENCH_WORKLOAD_KEY and while the source hash matches).
set -o pipefail
NAME=$1; KEY=$2; SHAPE=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAG=""
DEC="$GRAFT_REPO_ROOT/tools/bench_decode_attn.py $SHAPE --iters 6 --warmup 3 $FLAG"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $DEC > $OUT/kt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $DEC > $OUT/f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $DEC > $OUT/w.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
find $OUT/kt -name "*kernel_stats.csv" -exec cat {} \; | grep -i "decode_\|\"Name\"" > $OUT/kernel_stats.csv
{ echo "# $NAME: python3 tools/bench_decode_attn.py $SHAPE $FLAG  (bench workload: $KEY)"
  python tools/pmc_summary.py $OUT/f decode_mfma_ decode_merge_kernel
  python tools/pmc_summary.py $OUT/w decode_mfma_ decode_merge_kernel
  grep -h "^chunk" $OUT/f.log
  echo "# rocprofv3 --kernel-trace --stats (ns):"; cat $OUT/kernel_stats.csv; } > $OUT/summary.txt
NAME=$NAME KEY=$KEY SHAPE="$SHAPE $FLAG" python - <<'PY'
import json, os, re, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
name = os.environ["NAME"]
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_" + name)
vals, alg, attn = {}, None, "decode_mfma_kernel"
for line in open(os.path.join(out, "summary.txt")):
    m = re.match(r"(\S+)<.*?>\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=\s*([\d.]+)", line)
    if m:
        vals[(m.group(1), m.group(2))] = float(m.group(3))
        if m.group(1).startswith("decode_mfma_"):        # decode_mfma_range_kernel or decode_mfma_kernel: one of them runs
            attn = m.group(1)
    m = re.search(r"alg bytes ([\d.]+) GB", line)
    if m:
        alg = float(m.group(1)) * 1e9
g = lambda k, c: vals.get((k, c), 0.0)
hbm = (2 * g(attn, "FETCH_SIZE") + g(attn, "WRITE_SIZE")
       + g("decode_merge_kernel", "FETCH_SIZE") + g("decode_merge_kernel", "WRITE_SIZE")) * 1024.0
ns = {}
for line in open(os.path.join(out, "kernel_stats.csv")):
    m = re.match(r'"void sp::(decode_\w+)<.*?",(\d+),(\d+),([\d.]+)', line)
    if m:
        ns[m.group(1)] = float(m.group(4))
rec = {"workload": "tools/bench_decode_attn.py " + os.environ["SHAPE"], "bench_workload": os.environ["KEY"],
       "kernel": attn + "+decode_merge_kernel",
       "algorithmic_bytes": int(alg),
       "fetch_size_kib": g(attn, "FETCH_SIZE"), "write_size_kib": g(attn, "WRITE_SIZE"),
       "merge_fetch_kib": g("decode_merge_kernel", "FETCH_SIZE"), "merge_write_kib": g("decode_merge_kernel", "WRITE_SIZE"),
       "hbm_bytes_per_launch": int(hbm), "traffic_over_algorithmic": round(hbm / alg, 4),
       "rocprof_avg_ns": ns,
       "correction": "FETCH_SIZE x 2 for the 16-B/lane streaming gathers (gfx950), WRITE_SIZE and the merge as counted",
       "kernel_source_sha1": bench.decode_kernel_sources_sha1()}
json.dump(rec, open(os.path.join(out, f"decode_attn_pmc_{name}.json"), "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $OUT/f $OUT/w $OUT/kt
cat $OUT/summary.txt
