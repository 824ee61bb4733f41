#!/bin/bash
# SQ / TA / TCC counters of the Winograd stage's batched GEMM (tools/bench_conv.py "wino gemm"), round-3 kernel (RN_PERSIST=0) and the
# continuous-stream form (RN_PERSIST=1), one rocprofv3 pass per counter group.   bash tools/pmc_wino_gemm.sh OUTDIR
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_wino}
mkdir -p $out
for mode in 0 1; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA" \
             "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
    d=$out/m$mode/g$i
    rm -rf "$d"
    RN_PERSIST=$mode timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- python3 tools/bench_conv.py --only "wino gemm 36 x T 256->256" --mfma split --iters 5 > $out/m${mode}_g$i.log 2>&1 || echo "mode $mode group $i failed"
    i=$((i+1))
  done
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for mode in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob("%s/m%d/g*/**/*counter_collection.csv" % (out, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "mf16" in n:
                agg[n[:70] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("%s/m%d/g0/**/*kernel_trace.csv" % (out, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "mf16" in n:
                dur[n[:70] + " grid " + r.get("Grid_Size", "?")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("=== RN_PERSIST=%d" % mode)
    for k, d in agg.items():
        print(k, " launches", len(dur[k]), " avg %.1f us" % (sum(dur[k]) / max(len(dur[k]), 1)))
        for c, v in sorted(d.items()):
            print("   %-32s avg %.4g" % (c, sum(v) / len(v)))
PY
