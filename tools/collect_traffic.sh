#!/bin/bash
# HBM traffic of the bench step from the TCC counters, one rocprofv3 pass per counter (FETCH_SIZE takes 3 of the 4
# TCC slots, WRITE_SIZE 2: /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run on the GPU box:
#   bash tools/collect_traffic.sh            -> gpurun_out/traffic/{fetch,write}/...counter_collection.csv
# then  python tools/summarize_traffic.py gpurun_out/traffic > profiles/pmc_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/traffic/$c
  rm -rf "$d"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 bench.py --steps 1 --warmup 1 --sections headline --no-cpu-baseline --no-kernel-timing > gpurun_out/traffic_$c.log 2>&1
done
