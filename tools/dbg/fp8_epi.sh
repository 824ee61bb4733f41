mkdir -p gpurun_out/r6c
timeout -k 10 600 python -m pytest tests/test_gpu_conv_fp8_p8.py tests/test_gpu_conv_fp8.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r6c/fp8_tests.txt
grep -q "passed" gpurun_out/r6c/fp8_tests.txt && ! grep -q "failed" gpurun_out/r6c/fp8_tests.txt || exit 1
for rep in 1 2; do
for v in base product; do
  if [ $v = product ]; then unset RN_LIB_PATH; else export RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/lib$v.so; fi
  echo "== $v" | tee -a gpurun_out/r6c/fp8_micro.txt
  timeout -k 10 300 python tools/bench_conv_fp8.py 2>/dev/null | grep -v "^fp32" | tee -a gpurun_out/r6c/fp8_micro.txt
done
done
