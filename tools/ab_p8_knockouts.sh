#!/bin/bash
# Knock-outs of the eight-wave bf16 kernel (csrc/conv_bf16_p8.hip, -DP8_ABL=bits: 1 no epilogue, 2 no validity test, 4 no staging after the
# prologue) on the dominant layer shape; libraries built by hand into csrc/build_ab.  -> gpurun_out/p8/knockouts.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/p8
O=gpurun_out/p8/knockouts.txt
: > $O
for n in 0 1 2 4 5; do
  lib=3d-playground_amd/csrc/build_ab/lib_p8_abl$n.so
  [ $n = 0 ] && lib=3d-playground_amd/retinanet_mi355x/lib/libretinanet_mi355x.so
  echo "== P8_ABL=$n" >> $O
  RN_LIB_PATH=$PWD/$lib RN_BF16_P8=2 timeout -k 10 200 python3 tools/bench_conv_bf16.py --no-fp32 --only "head tower" 2>&1 | grep -E "fprop|dgrad" >> $O || exit 1
done
cat $O
