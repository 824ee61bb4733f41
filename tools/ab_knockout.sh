#!/bin/bash
# Knock-out timing of the SPLIT 3 implicit-GEMM kernel (wrong results, timing only): which part of a K-step is additive.
# K1 no B loads, K2 no A loads, K3 neither, K4 one MFMA of six, K7 all three.  Usage: tools/ab_knockout.sh OUTDIR
out=${1:-gpurun_out/ko}
mkdir -p $out
LIB=3d-playground_amd/retinanet_mi355x/lib
for v in D K1 K2 K3 K4 K7; do
  lib=$LIB/ab/lib$v.so; [ "$v" = "D" ] && lib=$LIB/libretinanet_mi355x.so
  echo "=== $v" | tee -a $out/ko.log
  for only in "head 3x3 256->256 P3" "l3 1x1 1024->256" "l4 1x1 512->2048"; do
    RN_LIB_PATH=$lib python tools/bench_conv.py --mfma split --iters 10 --only "$only" 2>&1 | grep -v "^fp32\|^layer\|amdgpu.ids" | tee -a $out/ko.log
  done
done
