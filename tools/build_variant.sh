#!/bin/bash
# A diagnostic / A-B build of the library beside the product one:  tools/build_variant.sh NAME "-DRN_KO=3 ..."
# (built with -DRN_EXPERIMENT=1: only such builds honour the knock-out / stamp / ablation macros, csrc/common.h)
#   -> 3d-playground_amd/retinanet_mi355x/lib/ab/libNAME.so   (use with RN_LIB_PATH=...; git-ignored like every built .so)
set -e
name=$1; shift
cd "$(dirname "$0")/../3d-playground_amd/csrc"
mkdir -p ../retinanet_mi355x/lib/ab
make -j8 OBJDIR=build_$name OUT=../retinanet_mi355x/lib/ab/lib$name.so \
     CFLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -DRN_EXPERIMENT=1 $*" 2>&1 | grep -i "error" || true
ls -la ../retinanet_mi355x/lib/ab/lib$name.so
