#!/bin/bash
# A/B of the split-operand implicit-GEMM variants on the GPU box (one call): conv parity tests on the default library, then
# tools/bench_conv.py on the dominant shapes for every (library, RN_SPLIT_A_ONCE) pair, then short training benches.
# Usage: tools/ab_split.sh OUTDIR
set -o pipefail
out=${1:-gpurun_out/ab}
mkdir -p $out
python -m pytest tests/test_gpu_conv.py -x -q > $out/conv_tests.log 2>&1 || { echo "conv tests FAILED"; tail -30 $out/conv_tests.log; exit 1; }
tail -2 $out/conv_tests.log
LIB=3d-playground_amd/retinanet_mi355x/lib
for cfg in "A 0" "A 1" "C 0" "C 1" "D 0" "D 1"; do
  set -- $cfg
  lib=$LIB/ab/lib$1.so; [ "$1" = "D" ] && lib=$LIB/libretinanet_mi355x.so
  echo "=== lib $1 (A: no pin, no sgb; C: pin; D: pin + sgb) RN_SPLIT_A_ONCE=$2" | tee -a $out/micro.log
  for only in "head 3x3 256->256 P3" "l3 1x1 1024->256" "l2 3x3 128->128" "l4 1x1 512->2048" "l2 1x1 128->512"; do
    RN_LIB_PATH=$lib RN_SPLIT_A_ONCE=$2 python tools/bench_conv.py --mfma split --iters 10 --only "$only" 2>&1 | grep -v "^fp32\|^layer" | tee -a $out/micro.log
  done
done
for cfg in "A 0" "D 0" "D 1"; do
  set -- $cfg
  lib=$LIB/ab/lib$1.so; [ "$1" = "D" ] && lib=$LIB/libretinanet_mi355x.so
  echo "=== step: lib $1 RN_SPLIT_A_ONCE=$2" | tee -a $out/step.log
  RN_LIB_PATH=$lib RN_SPLIT_A_ONCE=$2 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print(l['value'], l['ms_per_step'], {n:(k[n]['ms_per_step'],k[n]['frac']) for n in ('conv_igemm_2x2','conv_igemm_4x1','conv_wgrad','wino_input','wino_output') if n in k}, l.get('fp32_native_mfma',{}).get('value'))" | tee -a $out/step.log
done
