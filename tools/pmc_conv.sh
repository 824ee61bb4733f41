#!/bin/bash
# SQ counters of the conv kernels on one layer shape (default: head 3x3 256->256 at P3), one rocprofv3 pass per group.
#   bash tools/pmc_conv.sh ["layer name substring of tools/bench_conv.py"]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ONLY="${1:-head 3x3 256->256 P3}"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS"; do
  d=gpurun_out/pmc_conv/g$i
  rm -rf "$d"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- python3 tools/bench_conv.py --only "$ONLY" > gpurun_out/pmc_conv_g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_conv/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv_igemm" in n or "conv_wgrad" in n:
            agg[n[:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_conv/g0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv_igemm" in n or "conv_wgrad" in n:
            dur[n[:52]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(k, " launches", len(dur[k]), " avg %.1f us" % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c, v in sorted(d.items()):
        print("   %-30s avg %.4g" % (c, sum(v) / len(v)))
PY
