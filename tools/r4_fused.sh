#!/bin/bash
# Round 4: fused split merge - parity tests, then fused vs separate-merge timing at the headline and per-rank shapes
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4b}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fp8_kv.py tests/test_gpu_plan_overflow.py tests/test_gpu_vision.py tests/test_gpu_full_size.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
T=$GRAFT_REPO_ROOT/tools/bench_decode_attn.py
{
for f in "" "--no-fuse"; do
echo "== headline $f";   python3 $T --chunks 512 --iters 60 --warmup 20 $f
echo "== hkv1 $f";       python3 $T --bs 128 --Hq 8 --Hkv 1 --chunks 256,512 --iters 60 --warmup 20 $f
echo "== hkv2 $f";       python3 $T --bs 128 --Hq 16 --Hkv 2 --chunks 512 --iters 60 --warmup 20 $f
echo "== ctx1024 $f";    python3 $T --ctx 1024 --chunks 512 --iters 60 --warmup 20 $f
echo "== bs1 $f";        python3 $T --bs 1 --ctx 1024 --chunks 64 --iters 60 --warmup 20 $f
echo "== bs8 $f";        python3 $T --bs 8 --ctx 1024 --chunks 64 --iters 60 --warmup 20 $f
echo "== bs32 $f";       python3 $T --bs 32 --ctx 1024 --chunks 256 --iters 60 --warmup 20 $f
echo "== fp8 $f";        python3 $T --kv fp8 --chunks 512 --iters 60 --warmup 20 $f
done
} > $OUT/ab.txt 2>&1 || { tail -20 $OUT/ab.txt; exit 1; }
grep -v amdgpu.ids $OUT/ab.txt
