#!/bin/bash
# A/B of hand-built libraries (csrc/build_ab/*.so against the tree's) on the eight-wave bf16 weight gradient: bash tools/ab_wp8_libs.sh lib1 lib2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in tree "$@"; do
  lib=$PWD/3d-playground_amd/csrc/build_ab/$n.so
  [ $n = tree ] && lib=$PWD/3d-playground_amd/retinanet_mi355x/lib/libretinanet_mi355x.so
  echo "== $n"
  RN_LIB_PATH=$lib RN_BF16_P8=2 timeout -k 10 200 python3 tools/bench_conv_bf16.py --no-fp32 --only "3x3 256" 2>&1 | grep -E "wgrad" | cut -c1-84 || exit 1
done
