#!/bin/bash
# (round 5: the 256 x 256 tile this script compares with is quarantined beside it -- kept for the record of profiles/r03_split_knockout.txt)
# Knock-out timing of the split-operand implicit-GEMM kernels (wrong results, timing only): which part of a K-step is additive.
# K3 no loads after the prologue, K4 one MFMA of six, K8 no split arithmetic, K15 all of them; each with the 128 x 128 tile
# (RN_BIG_TILE=0) and the 256 x 256 tile (=1).  Usage: tools/ab_knockout.sh OUTDIR
out=${1:-gpurun_out/ko}
mkdir -p $out
LIB=3d-playground_amd/retinanet_mi355x/lib
for big in 0 1; do
for v in D K3 K4 K8 K15; do
  lib=$LIB/ab/lib$v.so; [ "$v" = "D" ] && lib=$LIB/libretinanet_mi355x.so
  echo "=== $v RN_BIG_TILE=$big" | tee -a $out/ko.log
  for only in "head 3x3 256->256 P3" "l3 1x1 1024->256"; do
    RN_BIG_TILE=$big RN_LIB_PATH=$lib python tools/bench_conv.py --mfma split --iters 10 --only "$only" 2>&1 | grep -v "^fp32\|^layer\|amdgpu.ids" | tee -a $out/ko.log
  done
done
done
