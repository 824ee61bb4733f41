#!/bin/bash
# SQ / TA / TCC counters of the eight-wave phased kernels (csrc/conv_bf16_p8.hip, conv_fp8_p8.hip) on the dominant layer shape (3x3 256->256
# at 135x240), one rocprofv3 pass per counter group and engine.   bash tools/pmc_p8.sh   -> gpurun_out/pmc_p8/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_p8
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  for eng in bf16 fp8; do
    d=gpurun_out/pmc_p8/${eng}_g$i
    rm -rf "$d"
    if [ $eng = bf16 ]; then cmd="tools/bench_conv_bf16.py --only head --no-fp32"; else cmd="tools/bench_conv_fp8.py --only head"; fi
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- python3 $cmd > gpurun_out/pmc_p8/${eng}_g$i.log 2>&1 || echo "group $i $eng failed"
  done
  i=$((i+1))
done
python3 - <<'PY' > gpurun_out/pmc_p8/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_p8/*_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "p8" in n:
            agg[n[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_p8/*_g0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "p8" in n:
            dur[n[:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(k, " launches", len(dur[k]), " avg %.1f us (under the counters)" % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c, v in sorted(d.items()):
        print("   %-32s avg %.4g" % (c, sum(v) / len(v)))
PY
cat gpurun_out/pmc_p8/summary.txt
