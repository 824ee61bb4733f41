#!/bin/bash
# SQ / TA / TCP counters of the split-operand fp32 conv kernels on one layer shape (default: head 3x3 256->256 at P3), one rocprofv3 pass
# per counter group.   bash tools/pmc_conv_split.sh ["layer name substring of tools/bench_conv.py"]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ONLY="${1:-head 3x3 256->256 P3}"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=gpurun_out/pmc_split/g$i
  rm -rf "$d"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- python3 tools/bench_conv.py --only "$ONLY" --mfma ${RN_PMC_MODE:-split} > gpurun_out/pmc_split_g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_split/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "split" in n or "wgrad" in n or "mf16" in n:
            agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_split/g0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "split" in n or "wgrad" in n or "mf16" in n:
            dur[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(k, " launches", len(dur[k]), " avg %.1f us" % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c, v in sorted(d.items()):
        print("   %-32s avg %.4g" % (c, sum(v) / len(v)))
PY
