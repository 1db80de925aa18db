#!/bin/bash
# Round 4: PMC + kernel-trace records of the decode attention at four workloads, the Hkv = 1 batch sweep
# (launch-size intercept / marginal bandwidth), and the extend persistent-form A/B in the prefill bench.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4d}
mkdir -p $OUT
ONLY=${2:-all}     # "pmc": the five PMC records only
if [ "$ONLY" = "all" ]; then
timeout -k 10 300 python -m pytest tests/test_gpu_plan_overflow.py tests/test_gpu_llama.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
fi
bash tools/pmc_decode.sh headline "llama3-8b|bs256|ctxuniform|kvauto" "--chunks 768 --interleave" nofuse > $OUT/pmc_headline.log 2>&1 &&
bash tools/pmc_decode.sh hkv1 "llama3-70b-tp8-rank|bs128|ctxuniform|kvauto" "--bs 128 --Hq 8 --Hkv 1 --chunks 768 --interleave" nofuse > $OUT/pmc_hkv1.log 2>&1 &&
bash tools/pmc_decode.sh ctx128 "llama3-8b|bs256|ctx128|kvauto" "--ctx 128 --chunks 128 --interleave" nofuse > $OUT/pmc_ctx128.log 2>&1 &&
bash tools/pmc_decode.sh bs32 "llama3-8b|bs32|ctx1024|kvauto" "--bs 32 --ctx 1024 --chunks 256 --interleave" nofuse > $OUT/pmc_bs32.log 2>&1 &&
bash tools/pmc_decode.sh fp8 "llama3-8b|bs256|ctxuniform|kvfp8_e5m2" "--kv fp8 --chunks 768 --interleave" nofuse > $OUT/pmc_fp8.log 2>&1 || { tail -5 $OUT/pmc_*.log; exit 1; }
for n in headline hkv1 ctx128 bs32 fp8; do tail -1 $OUT/pmc_$n.log | cut -c1-400; done
[ "$ONLY" = "pmc" ] && exit 0
T=$GRAFT_REPO_ROOT/tools/bench_decode_attn.py
{
echo "# Hkv = 1 (Hq 8, D 128, bf16, ctx U[128,4096], chunk 512, merge launch): time vs batch = launch size"
for bs in 32 64 128 256 512 1024; do echo "== bs $bs"; python3 $T --bs $bs --Hq 8 --Hkv 1 --chunks 512 --iters 40 --warmup 10 --no-fuse; done
echo "# Hkv = 8 headline heads, same sweep"
for bs in 16 32 64 128 256; do echo "== bs $bs"; python3 $T --bs $bs --chunks 512 --iters 40 --warmup 10 --no-fuse; done
} 2>&1 | grep -v amdgpu.ids > $OUT/sweep.txt
cat $OUT/sweep.txt
P="python3 bench.py --mode prefill --steps 2 --warmup 1"
{
echo "== prefill default (extend_w64_persist = 1: where it pays)"; timeout -k 10 400 $P 2>/dev/null | tail -1
echo "== prefill extend_w64_persist = 2 (every planned launch)"; SP_BENCH_DEBUG_SET="extend_w64_persist=2" timeout -k 10 400 $P 2>/dev/null | tail -1
echo "== prefill extend_w64_persist = 0"; SP_BENCH_DEBUG_SET="extend_w64_persist=0" timeout -k 10 400 $P 2>/dev/null | tail -1
echo "== prefill max-prefill-tokens 8192 (the reference's default chunk, server/args.py:33-34)"; timeout -k 10 400 $P --max-prefill-tokens 8192 2>/dev/null | tail -1
echo "== prefill max-prefill-tokens 65536"; timeout -k 10 400 $P --max-prefill-tokens 65536 2>/dev/null | tail -1
} > $OUT/prefill_ab.txt 2>&1
python3 - <<PY
import json
for line in open("$OUT/prefill_ab.txt"):
    if line.startswith("=="): print(line.strip()); continue
    try: d=json.loads(line)
    except Exception: print(line.strip()[:200]); continue
    r=d["roofline_prefill"]
    print("   ttft_p50", d["value"], "ms; pass", d["ms_per_step"], "ms; attention", r["achieved"], "TFLOP/s", r["avg_launch_ms"], "ms/launch x", r["launches"], "|", r["kernel"])
PY
