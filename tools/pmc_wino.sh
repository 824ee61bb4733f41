#!/bin/bash
# SQ / TA / TCP / TCC counters of the Winograd transforms on the head-tower group (tools/dbg/wino_stride.py runs the output and the input transform
# alone), one rocprofv3 pass per counter group.   bash tools/pmc_wino.sh OUTDIR -> OUTDIR/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_wino}
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum" "GRBM_GUI_ACTIVE"; do
  d=$out/g$i
  rm -rf "$d"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- python3 tools/dbg/wino_out_variants.py > $out/g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 - "$out" > "$out/summary.txt" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("%s/g*/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "wino" in n:
            agg[n[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("%s/g0/**/*kernel_trace.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "wino" in n:
            dur[n[:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(k, " launches", len(dur[k]), " avg %.1f us under the profiler" % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c, v in sorted(d.items()):
        print("   %-36s avg %.4g" % (c, sum(v) / len(v)))
PY
cat "$out/summary.txt"
