#!/bin/bash
# Full GPU check + the measurements quoted in DESIGN.md / BASELINE.md (run on the GPU box via gpurun).
set -o pipefail
mkdir -p gpurun_out/report
R=gpurun_out/report
# the whole report exceeds one gpurun call (20 min): run it as `stage1`, `stage2`, `--profile-only` (or with no argument on
# a box without that limit); every stage leaves its files in gpurun_out/report/
STAGE=${1:-all}
if [ "$STAGE" = "all" ] || [ "$STAGE" = "stage1" ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > $R/tests.log 2>&1 || { tail -30 $R/tests.log; exit 1; }
grep -h "passed\|failed" $R/tests.log | tail -1
grep -h "\[parity\] real width\|graph attention scratch\|100k-token\|auto-calibrated" $R/tests.log > $R/parity_lines.txt
timeout -k 10 400 python bench.py > $R/bench_decode.json 2> $R/bench_decode.err || { tail -20 $R/bench_decode.err; exit 1; }
tail -1 $R/bench_decode.json | cut -c1-400
timeout -k 10 400 python bench.py --mode prefill > $R/bench_prefill.json 2> $R/bench_prefill.err || { tail -20 $R/bench_prefill.err; exit 1; }
tail -1 $R/bench_prefill.json | cut -c1-300
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "stage2" ]; then
timeout -k 10 200 python tools/bench_extend_attn.py > $R/extend_attn.log 2>&1 || { tail -20 $R/extend_attn.log; exit 1; }
tail -8 $R/extend_attn.log
timeout -k 10 200 python tools/bench_sampling.py > $R/sampling.log 2>&1 || { tail -20 $R/sampling.log; exit 1; }
tail -8 $R/sampling.log
timeout -k 10 400 python bench.py --mode serve > $R/bench_serve.json 2> $R/bench_serve.err || { tail -20 $R/bench_serve.err; exit 1; }
tail -1 $R/bench_serve.json | cut -c1-300
timeout -k 10 400 python bench.py --mode serve --prefix 512 > $R/bench_serve_prefix.json 2> $R/bench_serve_prefix.err || { tail -20 $R/bench_serve_prefix.err; exit 1; }
for b in 1 8 32; do timeout -k 10 200 python bench.py --bs $b --ctx 1024 --no-cpu-baseline --no-ttft 2>/dev/null | tail -1 > $R/bench_bs$b.json || exit 1; cut -c1-140 $R/bench_bs$b.json; done
for c in 128 1024 4096; do timeout -k 10 300 python bench.py --ctx $c --no-cpu-baseline --no-ttft 2>/dev/null | tail -1 > $R/bench_ctx$c.json || exit 1; done
timeout -k 10 300 python bench.py --kv-cache-dtype fp8_e5m2 --no-cpu-baseline --no-ttft 2>/dev/null | tail -1 > $R/bench_fp8kv.json || exit 1
cut -c1-140 $R/bench_fp8kv.json
timeout -k 10 300 python bench.py --model llama3-70b-tp8-rank --bs 128 --no-cpu-baseline --no-ttft 2>/dev/null | tail -1 > $R/bench_70b_rank.json || exit 1
# config 4 rehearsals on this 1-GPU box (ranks share the GPU: gloo, or the direct IPC all-reduce) and the
# self-launching replica mode; the real thing needs the 8-GPU node: python bench.py --mode tp --gpus 8
timeout -k 10 300 python bench.py --rehearsal --mode tp --gpus 2 --tp 2 --layers 8 --steps 8 --warmup 2 2>/dev/null | tail -1 > $R/bench_tp2_rehearsal_gloo.json || exit 1
SP_CUSTOM_ALLREDUCE=1 timeout -k 10 300 python bench.py --rehearsal --mode tp --gpus 2 --tp 2 --layers 8 --steps 8 --warmup 2 2>/dev/null | tail -1 > $R/bench_tp2_rehearsal_direct.json || exit 1
timeout -k 10 300 python bench.py --rehearsal --gpus 2 --layers 8 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $R/bench_replicas2_rehearsal.json || exit 1
# replicas without --rehearsal on this 1-GPU box must be refused (exit code recorded), SP_LIBRARY_ROWS=auto must
# reproduce the table's substitutions, and the direct all-reduce vs its fusion with the norm (same-GPU rehearsal)
python bench.py --gpus 2 --layers 4 --steps 4 --warmup 1 --no-cpu-baseline --no-ttft > $R/bench_replicas2_refused.log 2>&1; echo "exit code $?" >> $R/bench_replicas2_refused.log
SP_LIBRARY_ROWS=auto timeout -k 10 300 python bench.py --no-cpu-baseline --no-ttft 2>/dev/null | tail -1 > $R/bench_librows_auto.json || exit 1
timeout -k 10 280 python tools/bench_allreduce.py --world 2 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" > $R/allreduce.log || exit 1
timeout -k 10 280 python tools/bench_allreduce.py --world 4 --blocks 0 32 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" >> $R/allreduce.log || exit 1
cut -c1-200 $R/bench_tp2_rehearsal_gloo.json
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "stage3" ]; then
# (the stamps twin of the library: bash tools/build_stamps.sh on the build host; skipped when it is not there)
if [ -f scratchpad_amd/lib/libscratchpad_hip_stamps.so ]; then
timeout -k 10 200 python tools/stamp_extend_attn.py --waves 8 > $R/extend_stamps.log 2>&1 || { tail -5 $R/extend_stamps.log; exit 1; }
fi
timeout -k 10 500 python tools/bench_mllama.py > $R/mllama.log 2>&1 || { tail -20 $R/mllama.log; exit 1; }
grep mllama $R/mllama.log
timeout -k 10 300 python tools/bench_gemv.py > $R/gemv.log 2>&1 || { tail -20 $R/gemv.log; exit 1; }
grep "per-layer" $R/gemv.log
# config 4 at FULL DEPTH as far as one GPU allows: Llama-3-70B, all 80 layers, bs 128, sharded TP = 4 over four ranks that
# share this GPU (140 GB of weights + the KV pools fit in 288 GB; TP = 8 would need 8 processes on the card, the box allows 6),
# direct all-reduce + fused norm: a rehearsal of the workload's code path (161 collectives per step), not of xGMI
SP_CUSTOM_ALLREDUCE=1 timeout -k 10 600 python bench.py --rehearsal --mode tp --gpus 4 --tp 4 --model llama3-70b --steps 6 --warmup 2 2>$R/bench_tp4_70b_full_depth.err | tail -1 > $R/bench_tp4_70b_full_depth.json || { tail -5 $R/bench_tp4_70b_full_depth.err; exit 1; }
cut -c1-300 $R/bench_tp4_70b_full_depth.json
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "--profile-only" ]; then
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-ttft > $GRAFT_REPO_ROOT/$R/prof_bench.json 2> $GRAFT_REPO_ROOT/$R/prof.err || { tail -20 $GRAFT_REPO_ROOT/$R/prof.err; exit 1; }
cd $GRAFT_REPO_ROOT
STATS=$(find $R/prof -name "*kernel_stats.csv" | head -1)
TRACE=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $STATS 22 > $R/kernel_stats.txt
python tools/prof_summary.py --steady $TRACE decode_mfma_ decode_merge_kernel >> $R/kernel_stats.txt
tail -1 $R/prof_bench.json | cut -c1-300 >> $R/kernel_stats.txt
rm -rf $R/prof
cat $R/kernel_stats.txt
# the prefill pass under the same profiler: the extend kernels' own durations (which of the two ran, and how long)
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/prof2 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode prefill --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$R/prof_prefill.json 2>$GRAFT_REPO_ROOT/$R/prof_prefill.err || { tail -5 $GRAFT_REPO_ROOT/$R/prof_prefill.err; exit 1; }
cd $GRAFT_REPO_ROOT
STATS=$(find $R/prof2 -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $STATS 12 > $R/prefill_kernel_stats.txt
tail -1 $R/prof_prefill.json | cut -c1-400 >> $R/prefill_kernel_stats.txt
rm -rf $R/prof2
cat $R/prefill_kernel_stats.txt
fi
