# A/B of the fp16-split kernel's activation prefetch depth (RN_MF16_PF): micro-benchmark + training step, variants interleaved on one box
mkdir -p gpurun_out/r6b
VARIANTS="product mfpf2 mfpf4" OUT=pf_micro bash tools/dbg/mf16_ko.sh > /dev/null
cat gpurun_out/r6b/pf_micro.txt
for rep in 1 2; do
for v in product mfpf2 mfpf4; do
  if [ $v = product ]; then unset RN_LIB_PATH; else export RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/lib$v.so; fi
  echo "== $v" | tee -a gpurun_out/r6b/pf_step.txt
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --sections headline --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print(j['value'], j['ms_per_step'], {n: k[n]['ms_per_step'] for n in k})" | tee -a gpurun_out/r6b/pf_step.txt
done
done
