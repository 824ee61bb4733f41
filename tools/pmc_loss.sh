#!/bin/bash
# SQ / TCC counters of the fused loss kernels on the microbenchmark (one rocprofv3 pass per counter group).
#   bash tools/pmc_loss.sh   -> gpurun_out/pmc_loss/<group>/...counter_collection.csv ; summary on stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/pmc_loss/g$i
  rm -rf "$d"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- python3 tools/bench_loss.py > gpurun_out/pmc_loss_g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_loss/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "focal" in n:
            agg[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s avg %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
