# A/B of one environment variable on the training step, interleaved on one box:  tools/dbg/ab_env.sh NAME VALUE_A VALUE_B [reps] [bench args]
name=$1; a=$2; b=$3; reps=${4:-2}; shift 4
mkdir -p gpurun_out/ab
for rep in $(seq $reps); do
for v in $a $b; do
  echo "== $name=$v" | tee -a gpurun_out/ab/$name.txt
  env $name=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --sections headline --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print(j['value'], j['ms_per_step'], {n: k[n]['ms_per_step'] for n in k})" | tee -a gpurun_out/ab/$name.txt
done
done
