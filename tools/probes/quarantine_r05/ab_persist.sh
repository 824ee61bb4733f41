#!/bin/bash
# A/B of round 4's continuous-stream plain GEMM (csrc/conv_igemm_mf16p.hip: persistent workgroups, two accumulator sets, the stores of
# tile n spread over the K-steps of tile n + 1) against the round-3 kernel it replaces for the Winograd stage (RN_PERSIST=0: the same
# library with the form switched off), one GPU box, one call, rounds interleaved.
#   configs: off = RN_PERSIST=0 | on = RN_PERSIST=1 (default) | on384 / on768 = the same on 384 / 768 workgroups (default 512 = 2 per CU)
# Usage: tools/ab_persist.sh OUTDIR [rounds]
set -o pipefail
out=${1:-gpurun_out/abp}; rounds=${2:-2}
mkdir -p $out
LIB=3d-playground_amd/retinanet_mi355x/lib
run() {   # name -> env
  case $1 in
    off)   echo "RN_PERSIST=0";;
    on)    echo "RN_PERSIST=1";;
    on384) echo "RN_PERSIST=1 RN_PERSIST_WGS=384";;
    on768) echo "RN_PERSIST=1 RN_PERSIST_WGS=768";;
  esac
}
for r in $(seq 1 $rounds); do
  for cfg in off on on384 on768; do
    echo "=== round $r $cfg" | tee -a $out/micro.log
    for only in "wino gemm"; do
      env $(run $cfg) python tools/bench_conv.py --mfma split --iters 20 --only "$only" 2>&1 | grep -v "^fp32\|^layer\|amdgpu.ids" | tee -a $out/micro.log
    done
  done
done
for r in $(seq 1 $rounds); do
  for cfg in off on; do
    echo "=== step round $r $cfg" | tee -a $out/step.log
    env $(run $cfg) python bench.py --steps 8 --warmup 3 --sections headline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print(l['value'], l['ms_per_step'], {n:(k[n]['ms_per_step'],k[n]['frac']) for n in ('conv_igemm_2x2','conv_igemm_4x1','conv_wgrad','wino_input','wino_output') if n in k})" | tee -a $out/step.log
  done
done
