mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_full_size.py tests/test_gpu_fp8_kv.py tests/test_gpu_long_context.py -x -q > gpurun_out/r5/tests_st.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r5/tests_st.log
if [ $rc -ne 0 ]; then exit $rc; fi
L="libscratchpad_hip_prev.so libscratchpad_hip.so"
for shape in headline bs64 bs128 ctx1024 ctx4096 hkv1; do
  timeout -k 10 120 python tools/ab_decode.py $L --shape $shape 2>&1 | grep -v amdgpu.ids || exit 1
done
timeout -k 10 120 python tools/ab_decode.py $L --shape headline --graph-slots 2>&1 | grep -v amdgpu.ids || exit 1
timeout -k 10 120 python tools/ab_decode.py $L --shape headline --kv fp8 2>&1 | grep -v amdgpu.ids || exit 1
timeout -k 10 120 python tools/ab_decode.py $L --shape bs8 --chunk 128 2>&1 | grep -v amdgpu.ids || exit 1
