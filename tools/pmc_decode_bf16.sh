#!/bin/bash
# PMC traffic of the bf16 decode attention kernel at the headline shapes: FETCH_SIZE and WRITE_SIZE in
# separate passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots), then the traffic / algorithmic ratio
# and the hash of the kernel sources it was measured on -> gpurun_out/pmc2/{summary.txt,decode_attn_pmc.json}
# (copy them to profiles/rNN_decode_attn_pmc.{txt,json}; bench.py uses the ratio only while the hash matches)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DEC="$GRAFT_REPO_ROOT/tools/bench_decode_attn.py --chunks 512 --iters 4 --warmup 2"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $DEC > $OUT/f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $DEC > $OUT/w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/f decode_mfma_kernel decode_merge_kernel > $OUT/summary.txt
python tools/pmc_summary.py $OUT/w decode_mfma_kernel decode_merge_kernel >> $OUT/summary.txt
grep -h chunk $OUT/f.log >> $OUT/summary.txt
python - <<'PY'
import json, os, re, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc2")
vals = {}
for line in open(os.path.join(out, "summary.txt")):
    m = re.match(r"(\S+)<.*?>\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=\s*([\d.]+)", line)
    if m:
        vals[(m.group(1), m.group(2))] = float(m.group(3))
    m = re.search(r"alg bytes ([\d.]+) GB", line)
    if m:
        alg = float(m.group(1)) * 1e9
kib = 1024.0
hbm = (2 * vals[("decode_mfma_kernel", "FETCH_SIZE")] + vals[("decode_mfma_kernel", "WRITE_SIZE")]
       + vals[("decode_merge_kernel", "FETCH_SIZE")] + vals[("decode_merge_kernel", "WRITE_SIZE")]) * kib
rec = {"workload": "bs=256 Hq=32 Hkv=8 D=128 bf16 ctx=U[128,4096] seed 0 chunk=512",
       "kernel": "decode_mfma_kernel+decode_merge_kernel", "algorithmic_bytes": int(alg),
       "fetch_size_kib": vals[("decode_mfma_kernel", "FETCH_SIZE")], "write_size_kib": vals[("decode_mfma_kernel", "WRITE_SIZE")],
       "merge_fetch_kib": vals[("decode_merge_kernel", "FETCH_SIZE")], "merge_write_kib": vals[("decode_merge_kernel", "WRITE_SIZE")],
       "hbm_bytes_per_launch": int(hbm), "traffic_over_algorithmic": round(hbm / alg, 4),
       "correction": "FETCH_SIZE x 2 for the 16-B/lane streaming gathers (gfx950), WRITE_SIZE and the merge as counted",
       "kernel_source_sha1": bench.decode_kernel_sources_sha1()}
json.dump(rec, open(os.path.join(out, "decode_attn_pmc.json"), "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $OUT/f $OUT/w
cat $OUT/summary.txt
