#!/bin/bash
# SQ / TA / TCP / TCC counters of the fp16-split GEMM on the Winograd stage's launch (tools/bench_conv.py "wino gemm", split3): the register form of
# its activation loads (RN_MF16_STG_MIN_K=100000) against the staged form (default), one rocprofv3 pass per counter group.
#   bash tools/pmc_mf16_stg.sh OUTDIR     -> OUTDIR/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_stg}
mkdir -p $out
for mode in reg stg; do
  i=0
  if [ $mode = reg ]; then export RN_MF16_STG_MIN_K=100000; else unset RN_MF16_STG_MIN_K; fi
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT" \
             "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
    d=$out/$mode/g$i
    rm -rf "$d"
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- python3 tools/bench_conv.py --only "wino gemm 36 x T 256->256" --mfma split3 --iters 5 > $out/${mode}_g$i.log 2>&1 || echo "mode $mode group $i failed"
    i=$((i+1))
  done
done
python3 - "$out" > "$out/summary.txt" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for mode in ("reg", "stg"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob("%s/%s/g*/**/*counter_collection.csv" % (out, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "mf16" in n:
                agg[n[:60] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("%s/%s/g0/**/*kernel_trace.csv" % (out, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "mf16" in n:
                dur[n[:60] + " grid " + r.get("Grid_Size", "?")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("=== activation loads: %s" % ("straight into registers (RN_MF16_STG_MIN_K=100000)" if mode == "reg" else "staged through the LDS (default)"))
    for k, d in agg.items():
        print(k, " launches", len(dur[k]), " avg %.1f us under the profiler" % (sum(dur[k]) / max(len(dur[k]), 1)))
        for c, v in sorted(d.items()):
            print("   %-36s avg %.4g" % (c, sum(v) / len(v)))
PY
cat "$out/summary.txt"
