set -e
mkdir -p gpurun_out/r6b; out=gpurun_out/r6b/${OUT:-ko}.txt
: > $out
for v in ${VARIANTS:-product mfko1 mfko2 mfko4 mfko5 mfko6 mfko8}; do
  if [ $v = product ]; then unset RN_LIB_PATH; else export RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/lib$v.so; fi
  echo "== $v" >> $out
  for only in "wino gemm 36 x T 256" "l1 1x1" "l3 1x1" "l2 1x1"; do
    timeout -k 10 120 python tools/bench_conv.py --mfma split3 --iters 20 --only "$only" 2>/dev/null | grep -v "^fp32\|^layer" >> $out
  done
done
cat $out
