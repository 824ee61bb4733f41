# A/B of library builds on the training step, interleaved on one box:  tools/dbg/ab_lib.sh "name[:ENV=VAL] ..." [reps]
reps=${2:-2}
mkdir -p gpurun_out/ab
for rep in $(seq $reps); do
for spec in $1; do
  v=${spec%%:*}; envs=""
  [ "$spec" != "$v" ] && envs=$(echo ${spec#*:} | tr ':' ' ')
  if [ $v = product ]; then lib=""; else lib="RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/lib$v.so"; fi
  echo "== $spec" | tee -a gpurun_out/ab/libs.txt
  env $lib $envs timeout -k 10 300 python bench.py --steps 10 --warmup 3 --sections headline --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print(j['value'], j['ms_per_step'], {n: k[n]['ms_per_step'] for n in k})" | tee -a gpurun_out/ab/libs.txt
done
done
