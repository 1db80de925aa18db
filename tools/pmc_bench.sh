#!/bin/bash
# PMC traffic of the decode attention kernels INSIDE the model: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate
# passes) around `python3 bench.py` itself (HIP-graph replay, the headline workload unless arguments are given) ->
# gpurun_out/pmc_bench/{summary.txt,bench_pmc.json}; copy to profiles/rNN_bench_pmc.{txt,json}.
# bench.py:pmc_traffic prefers this in-model record (same kernel-source hash, same workload key) over the kernel-alone one.
#   bash tools/pmc_bench.sh [KEY] [bench.py arguments ...]
set -o pipefail
KEY=${1:-"llama3-8b|bs256|ctxuniform|kvauto"}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ttft --profile-steps 1 $*"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $B > $OUT/f.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $B > $OUT/w.log 2>&1 || { tail -5 $OUT/f.log $OUT/w.log; exit 1; }
cd $GRAFT_REPO_ROOT
{ echo "# rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE} --kernel-trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ttft --profile-steps 1 $*"
  python tools/pmc_summary.py $OUT/f decode_mfma_ decode_merge_kernel
  python tools/pmc_summary.py $OUT/w decode_mfma_ decode_merge_kernel
  grep -h '^{"metric"' $OUT/f.log | tail -1 | cut -c1-600; } > $OUT/summary.txt
KEY=$KEY python - <<'PY'
import csv, glob, json, os, re, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_bench")
def steady(root, counter):
    """per-launch counter values of the model's steady-state launches: the decode_mfma launches above half the largest
    value (drops graph-capture warm-ups at the padded fill length) and all merge launches behind them"""
    vals = {"decode_mfma_": [], "decode_merge_kernel": []}     # decode_mfma_range_kernel / decode_mfma_kernel
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            for k in vals:
                if k in r["Kernel_Name"]:
                    vals[k].append(float(r["Counter_Value"]))
    big = max(vals["decode_mfma_"])
    a = [v for v in vals["decode_mfma_"] if v > 0.5 * big]
    m = vals["decode_merge_kernel"]
    mbig = max(m) if m else 0.0
    mm = [v for v in m if v > 0.5 * mbig] if m else [0.0]
    return sum(a) / len(a), sum(mm) / len(mm), len(a)
fa, fm, n = steady(os.path.join(out, "f"), "FETCH_SIZE")
wa, wm, _ = steady(os.path.join(out, "w"), "WRITE_SIZE")
line = json.loads([l for l in open(os.path.join(out, "f.log")) if l.startswith('{"metric"')][-1])
alg = line["roofline"]["algorithmic_bytes_per_launch"]
hbm = (2 * fa + wa + fm + wm) * 1024.0
rec = {"workload": "python3 bench.py " + " ".join(sys.argv[1:]) + " (in the model, HIP-graph replay)", "bench_workload": os.environ["KEY"],
       "kernel": line["roofline"]["kernel"], "algorithmic_bytes": int(alg), "launches_averaged": n,
       "fetch_size_kib": fa, "write_size_kib": wa, "merge_fetch_kib": fm, "merge_write_kib": wm,
       "hbm_bytes_per_launch": int(hbm), "traffic_over_algorithmic": round(hbm / alg, 4), "in_model": True,
       "correction": "FETCH_SIZE x 2 for the 16-B/lane streaming gathers (gfx950), WRITE_SIZE and the merge as counted",
       "kernel_source_sha1": bench.decode_kernel_sources_sha1()}
json.dump(rec, open(os.path.join(out, "bench_pmc.json"), "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $OUT/f $OUT/w
cat $OUT/summary.txt | cut -c1-300
