mkdir -p gpurun_out/r4g
python -m pytest tests/test_gpu_conv_big.py tests/test_gpu_conv.py -x -q > gpurun_out/r4g/conv_tests.log 2>&1; tail -2 gpurun_out/r4g/conv_tests.log
grep -q " passed" gpurun_out/r4g/conv_tests.log && ! grep -q "failed" gpurun_out/r4g/conv_tests.log || exit 1
for r in 1 2; do for lib in preT new; do
  if [ $lib = preT ]; then export RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/libpreT.so; else unset RN_LIB_PATH; fi
  echo "== $lib" >> gpurun_out/r4g/micro.log
  for only in "wino gemm 36 x T" "gemm-like" "ksweep 1x1 64->" "ksweep 1x1 512->" "l1 1x1 64->256" "l3 1x1 1024->256" "l2 1x1 128->512" "l4 1x1 512->2048" "head 3x3 256->256 P3" "l3 3x3 256->256"; do
    python tools/bench_conv.py --mfma split --iters 20 --only "$only" 2>&1 | grep -v "^fp32\|^layer\|amdgpu.ids" >> gpurun_out/r4g/micro.log
  done
done; done
for r in 1 2; do for lib in preT new; do
  if [ $lib = preT ]; then export RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/libpreT.so; else unset RN_LIB_PATH; fi
  python bench.py --steps 10 --warmup 3 --sections headline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('$lib', l['value'], l['ms_per_step'], {n:(k[n]['ms_per_step'],k[n]['frac']) for n in ('conv_igemm_2x2','conv_igemm_4x1','conv_wgrad','wino_input','wino_output') if n in k})" >> gpurun_out/r4g/step.log
done; done
cut -c1-100 gpurun_out/r4g/micro.log; cat gpurun_out/r4g/step.log
