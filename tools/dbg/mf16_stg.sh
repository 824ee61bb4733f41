mkdir -p gpurun_out/r6b
timeout -k 10 600 python -m pytest tests/test_gpu_conv_mf16.py tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r6b/stg_tests.txt
grep -q "passed" gpurun_out/r6b/stg_tests.txt && ! grep -q "failed" gpurun_out/r6b/stg_tests.txt || exit 1
VARIANTS="mfstg0 product" OUT=stg_micro bash tools/dbg/mf16_ko.sh > /dev/null
cat gpurun_out/r6b/stg_micro.txt
for rep in 1 2; do
for v in mfstg0 product; do
  if [ $v = product ]; then unset RN_LIB_PATH; else export RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/lib$v.so; fi
  echo "== $v" | tee -a gpurun_out/r6b/stg_step.txt
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --sections headline --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print(j['value'], j['ms_per_step'], {n: k[n]['ms_per_step'] for n in k})" | tee -a gpurun_out/r6b/stg_step.txt
done
done
