#!/bin/bash
# A/B of hand-built libraries (csrc/build_ab/*.so against the tree's) on the fp8 layer shapes: bash tools/ab_q8_libs.sh lib1 lib2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in tree "$@"; do
  lib=$PWD/3d-playground_amd/csrc/build_ab/$n.so
  [ $n = tree ] && lib=$PWD/3d-playground_amd/retinanet_mi355x/lib/libretinanet_mi355x.so
  echo "== $n"
  RN_LIB_PATH=$lib timeout -k 10 200 python3 tools/bench_conv_fp8.py 2>&1 | grep -v amdgpu.ids || exit 1
done
